/* crane_gpu/preempt.h — preemption inputs / outputs of the node-selection cycle.
 *
 * Replaces, on the reference side:
 *   - g_config.Preempt.PreemptType != PREEMPT_NONE and the per-QoS preempt lists NodeSelect reads
 *     (src/CraneCtld/JobScheduler.cpp:6522-6543: qos_preempt_map[job->qos] <- Qos::preempt);
 *   - the fields LocalScheduler::TryPreempt_ reads of every job (JobScheduler.cpp:6378-6505):
 *     PdJobInScheduler {qos, qos_priority, priority} (JobScheduler.h:113-134),
 *     RnJobInScheduler {job_id, qos, qos_priority, start_time, end_time, allocated_res} (JobScheduler.h:56-70);
 *   - SchedulerAlgo::m_preempting_set_ (JobScheduler.h:984), which lives across cycles
 *     (JobScheduler.cpp:6545-6559: ids no longer running are dropped, the others end at now + 1 s);
 *   - the results: PdJobInScheduler::preempted_jobs (JobScheduler.h:124-126), reason "Preempted" on a pending job
 *     placed earlier in the same cycle (JobScheduler.cpp:6781-6784), g_job_scheduler->EnqueuePreemptCancel (:6793).
 * QoS names are dense ids (the caller's string table); job references in the results are indices into the cycle's
 * pending queue (bit 31 set) or running table (bit 31 clear).
 *
 * How it runs: a cycle with `enabled` set goes through k_select with every job on its general path and the device form
 * of TryPreempt_ / PreemptSegTree between the res_total selection and the backfill (csrc/preempt_dev.inc; DESIGN.md
 * 5j); bit-exact against the CPU oracle's restatement (tests/test_preempt.py).  Partitions that share nodes
 * and reservations are served together with it.  Candidates that
 * the reference's comparator leaves unordered (it sorts the iteration order of a hash set) are taken in ascending
 * index.  With `enabled == 0` the call is cns_select.
 *
 * Limits (exceeding one returns CNS_ERR_UNSUPPORTED with a message, never a device fault; the integrator then runs the
 * CPU SchedulerAlgo for the cycle): per partition, candidate / chosen lists of max(4096, running + pending jobs)
 * entries (capped so that all partitions together stay below 64 Mi entries) and a segment-tree pool of 65 536 nodes
 * (one job's trees: about node_num x (entries of the nodes' time maps inside its window) x log2 of that).
 */
#ifndef CRANE_GPU_PREEMPT_H
#define CRANE_GPU_PREEMPT_H

#include "node_select.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CNS_REASON_PREEMPTED 7          /* "Preempted" (JobScheduler.cpp:6783) */
#define CNS_PREEMPT_REF_PENDING 0x80000000u

typedef struct cns_preempt_soa {
  uint32_t enabled;                     /* g_config.Preempt.PreemptType != PREEMPT_NONE (PREEMPT_QOS)   */
  uint32_t num_qos;
  const uint32_t* qos_preempt_offsets;  /* [num_qos + 1] CSR: Qos::preempt of qos q ...                 */
  const uint32_t* qos_preempt;          /* ... as qos ids, in the QoS record's order                    */
  /* pending jobs, [cns_job_soa::num_jobs], queue order */
  const uint32_t* pd_job_id;
  const uint32_t* pd_qos;
  const uint32_t* pd_qos_priority;
  const double* pd_priority;            /* PdJobInScheduler::priority (JobScheduler.h:117)               */
  /* running jobs, [cns_running_soa::num_jobs] */
  const uint32_t* rn_job_id;
  const uint32_t* rn_qos;
  const uint32_t* rn_qos_priority;
  const int64_t* rn_start_sec;
  /* m_preempting_set_ as the previous cycle left it */
  uint32_t num_preempting;
  uint32_t reserved0;
  const uint32_t* preempting_job_ids;
} cns_preempt_soa;

typedef struct cns_preempt_out {
  uint64_t capacity;                    /* entries available in preempted[]                              */
  uint64_t* offsets;                    /* [num_jobs + 1] preempted_jobs of pending job j                */
  uint32_t* preempted;                  /* job references, in the reference's push_back order            */
  uint32_t cancel_capacity;             /* entries available in cancelled_job_ids[] (>= running jobs)    */
  uint32_t num_cancelled;               /* out                                                           */
  uint32_t* cancelled_job_ids;          /* EnqueuePreemptCancel, in call order                           */
  uint32_t preempting_capacity;         /* entries available in preempting_job_ids[]                     */
  uint32_t num_preempting;              /* out: m_preempting_set_ after the cycle (ascending)            */
  uint32_t* preempting_job_ids;
} cns_preempt_out;

/* cns_select with preemption; preempt == NULL or preempt->enabled == 0: exactly cns_select (pout may be NULL). */
int cns_select_preempt(cns_handle* h, int64_t now_sec, const cns_job_soa* jobs, const cns_preempt_soa* preempt,
                       cns_placement_soa* out, cns_preempt_out* pout);

#ifdef __cplusplus
}
#endif
#endif
