/* C ABI of the MI355X MultiFactorPriority sorter — the IPrioritySorter that SchedulerAlgo::NodeSelect
 * calls before its ordered loop when PriorityType is multifactor.
 *
 * Reference (paths relative to the CraneSched tree):
 *   class MultiFactorPriority : IPrioritySorter            src/CraneCtld/JobScheduler.h:203-231
 *   GetOrderedJobPtrVec / CalculateFactorBound_ / CalculatePriority_
 *                                                           src/CraneCtld/JobScheduler.cpp:7606-7819
 *   call site inside NodeSelect                             src/CraneCtld/JobScheduler.cpp:6735
 *   PriorityConfig {FavorSmall, MaxAge, Weight*}            src/CraneCtld/CtldPublicDefs.h:162-174
 *   selection of the sorter                                 src/CraneCtld/JobScheduler.cpp:150-156
 *
 * What it computes: min/max bounds of six job attributes over pending + running jobs, a per-account
 * "service value" accumulated over the running jobs IN VECTOR ORDER (fp64), one fp64 priority per pending job
 * (five weighted factors), and the pending jobs ordered by descending priority; jobs past `limit` are the
 * caller's "Priority" rejects (JobScheduler.cpp:7625-7630).
 *
 * Canonicalisation: the reference sorts with std::ranges::sort (unstable; ties in unspecified order).  Here
 * ties keep ascending input index (a stable sort), as SURVEY.md §8(f)-2 prescribes.  RnJobInScheduler::node_num
 * is never initialised in the reference (JobScheduler.h:70,76-89) although :7694 reads it; the caller passes
 * the number of allocated nodes.
 *
 * All arithmetic is integer or IEEE fp64 in the reference's operation order (-ffp-contract=off): priorities
 * are compared as bit patterns in the parity tests.  No CPU fallback: CNS_ERR_NO_DEVICE without a GPU.
 */
#ifndef CRANE_GPU_PRIORITY_H_
#define CRANE_GPU_PRIORITY_H_

#include <stdint.h>

#include "node_select.h"

#ifdef __cplusplus
extern "C" {
#endif

/* g_config.PriorityConfig, CtldPublicDefs.h:162-174 */
typedef struct cns_priority_config {
  uint64_t max_age_sec;        /* MaxAge                                   */
  uint32_t weight_age;         /* WeightAge                                */
  uint32_t weight_fair_share;  /* WeightFairShare                          */
  uint32_t weight_job_size;    /* WeightJobSize                            */
  uint32_t weight_partition;   /* WeightPartition                          */
  uint32_t weight_qos;         /* WeightQoS                                */
  uint32_t favor_small;        /* FavorSmall (bool)                        */
} cns_priority_config;

/* Pending jobs, input order = the vector handed to GetOrderedJobPtrVec.  Fields of PdJobInScheduler read at
 * JobScheduler.cpp:7664-7690 and :7759-7767. */
typedef struct cns_prio_pending_soa {
  uint32_t num_jobs;
  const int64_t* submit_sec;          /* [J] job->submit_time                                      */
  const uint32_t* qos_priority;       /* [J]                                                       */
  const uint32_t* partition_priority; /* [J]                                                       */
  const uint32_t* node_num;           /* [J]                                                       */
  const int64_t* total_cpu_raw;       /* [J] req_total_res_view cpu, raw = value * 256 (cpp:7156)  */
  const uint64_t* total_mem;          /* [J] req_total_res_view memory bytes                       */
  const uint32_t* account;            /* [J] dense account id < num_accounts                       */
  const double* cached_priority;      /* [J] priority kept from an earlier cycle; 0.0 = compute (cpp:7616); NULL = all 0.0 */
} cns_prio_pending_soa;

/* Running jobs, vector order (the fp64 service values are accumulated in this order, cpp:7716-7746). */
typedef struct cns_prio_running_soa {
  uint32_t num_jobs;
  const int64_t* start_sec;           /* [R] job->start_time                                       */
  const uint32_t* qos_priority;       /* [R]                                                       */
  const uint32_t* partition_priority; /* [R]                                                       */
  const uint32_t* node_num;           /* [R] number of allocated nodes                             */
  const int64_t* alloc_cpu_raw;       /* [R] allocated_res_view cpu, raw                           */
  const uint64_t* alloc_mem;          /* [R] allocated_res_view memory bytes                       */
  const uint32_t* account;            /* [R]                                                       */
} cns_prio_running_soa;

/* Orders the pending jobs.  order_out[i] = input index of the i-th job in descending priority (ties:
 * ascending input index), i < num_jobs; priority_out[j] = priority of input job j (the cached value when it
 * was non-zero); *num_ordered = min(num_jobs, limit) — entries order_out[*num_ordered ..] are the jobs the
 * reference marks "Priority".  rn may be NULL (no running jobs).  Synchronous. */
int cns_priority_order(cns_handle* h, int64_t now_sec, const cns_priority_config* cfg, uint32_t num_accounts,
                       const cns_prio_pending_soa* pd, const cns_prio_running_soa* rn, uint64_t limit,
                       uint32_t* order_out, double* priority_out, uint64_t* num_ordered);

/* HIP-event time of the device work of the last cns_priority_order (bounds + service values + priorities +
 * radix sort), and the bytes it moved through HBM by construction (for the roofline figure). */
int cns_priority_timing(const cns_handle* h, double* kernels_ms, uint64_t* algorithmic_bytes);

#ifdef __cplusplus
}
#endif
#endif /* CRANE_GPU_PRIORITY_H_ */
