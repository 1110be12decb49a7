/* C ABI of the MI355X step scheduler — JobInCtld::SchedulePendingSteps for every job that has pending steps, in
 * one call ("reuses the same feasibility / top-k logic within a job's allocation").
 *
 * Reference (paths relative to the CraneSched tree):
 *   JobInCtld::SchedulePendingSteps                      src/CraneCtld/CtldPublicDefs.cpp:2038-2159
 *   ResourceView::GetFeasibleResourceInNode              src/Utilities/PublicHeader/PublicHeader.cpp:519-599
 *   ResourceInNodeV3 -=                                  src/Utilities/PublicHeader/PublicHeader.cpp:789-796
 *
 * What it computes, per job, on the job's own allocation (`step_res_avail_`): the pending steps in FIFO order; for a
 * step the job's nodes are walked, a node qualifies with `ntasks_on_node` = how many task requests fit after the
 * per-node request (capped at ntasks_per_node_max, at least ntasks_per_node_min), the `node_num` nodes with the most
 * tasks are kept in a std::priority_queue (walk stops as soon as node_num nodes hold >= ntasks tasks), tasks are handed
 * out in the queue's pop order, every task getting its own GetFeasibleResourceInNode allocation, and the allocations
 * are taken out of step_res_avail_.  The first step that does not fit stops that job's queue (:2104-2106).
 * Jobs are independent of each other: one GPU thread per job.
 *
 * Canonicalisation: the reference walks `step_res_avail_.EachNodeResMap()`, an unordered_map (order unspecified); here
 * a job's nodes are walked in the order given (ascending dense node index).  Which of several equal-capacity nodes
 * leaves the queue, and the pop order, follow libstdc++'s heap exactly (as for NodeSelect's top-k queues).
 *
 * Limits: at most CNS_STEP_MAX_NODES nodes per step; the GRES layout is the handle's (cns_set_nodes).
 * No CPU fallback: CNS_ERR_NO_DEVICE without a GPU.
 */
#ifndef CRANE_GPU_STEPS_H_
#define CRANE_GPU_STEPS_H_

#include <stdint.h>

#include "node_select.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CNS_STEP_MAX_NODES 64u

/* Jobs with pending steps: their nodes with what is still free inside the job's allocation. */
typedef struct cns_step_job_soa {
  uint32_t num_jobs;
  uint32_t num_nodes;              /* = node_offsets[num_jobs]                                             */
  const uint32_t* node_offsets;    /* [num_jobs+1] CSR over the arrays below                               */
  const uint32_t* node_idx;        /* dense node index (matched against a step's include / exclude lists)  */
  const int64_t* avail_cpu_raw;    /* step_res_avail_.At(node): cpu raw                                    */
  const uint64_t* avail_mem;
  const uint64_t* avail_core_lo;
  const uint64_t* avail_core_hi;   /* may be NULL = 0 */
  const uint64_t* avail_gres;      /* may be NULL = 0 */
  const uint32_t* step_offsets;    /* [num_jobs+1] CSR over cns_step_soa: pending_step_ids_ in queue order */
} cns_step_job_soa;

/* Pending steps (CommonStepInCtld fields read at CtldPublicDefs.cpp:2052-2125), grouped by job. */
typedef struct cns_step_soa {
  uint32_t num_steps;
  const int64_t* node_cpu_raw;     /* req_node_res_view; NULL = 0 */
  const uint64_t* node_mem;        /* NULL = 0 */
  const uint8_t* node_gres_total;  /* [S][CNS_MAX_GRES_NAMES]; NULL = none */
  const uint8_t* node_gres_spec;   /* [S][CNS_MAX_GRES_CLASSES]; NULL = none */
  const int64_t* task_cpu_raw;     /* req_task_res_view */
  const uint64_t* task_mem;
  const uint8_t* task_gres_total;  /* NULL = none */
  const uint8_t* task_gres_spec;   /* NULL = none */
  const uint32_t* node_num;        /* 1 .. CNS_STEP_MAX_NODES */
  const uint32_t* ntasks;          /* >= node_num */
  const uint32_t* ntasks_per_node_min;
  const uint32_t* ntasks_per_node_max;
  const uint32_t* incl_offsets;    /* [S+1] CSR included_nodes (dense node indices); NULL = none */
  const uint32_t* incl_nodes;
  const uint32_t* excl_offsets;    /* [S+1] CSR excluded_nodes; NULL = none */
  const uint32_t* excl_nodes;
} cns_step_soa;

/* Results, caller-allocated.  Step s owns node records [place_offsets[s], place_offsets[s+1]) (prefix sum of
 * node_num) in the queue's pop order, and task records [task_offsets[s], task_offsets[s+1]) (prefix sum of ntasks):
 * task ids 0, 1, ... of the step in the order they were handed out (craned_task_map / task_res_map). */
typedef struct cns_step_result_soa {
  uint8_t* scheduled;        /* [S] 1: scheduled in this pass; 0: still pending (and so is every later step of its job) */
  uint64_t* place_offsets;   /* [S+1] filled by the engine */
  uint32_t* node_idx;        /* [sum node_num] CNS_NODE_NONE when not scheduled                      */
  uint32_t* node_ntasks;     /* tasks on the node                                                    */
  int64_t* node_cpu_raw;     /* step_alloc_res per node = per-node request + its tasks               */
  uint64_t* node_mem;
  uint64_t* node_core_lo;
  uint64_t* node_core_hi;
  uint64_t* node_gres;
  uint64_t* task_offsets;    /* [S+1] filled by the engine */
  uint32_t* task_node;       /* [sum ntasks] dense node index of the task                            */
  int64_t* task_cpu_raw;     /* task_res_map[task]                                                   */
  uint64_t* task_mem;
  uint64_t* task_core_lo;
  uint64_t* task_core_hi;
  uint64_t* task_gres;
  /* step_res_avail_ after the pass, same shape as the input node arrays */
  int64_t* avail_cpu_raw;
  uint64_t* avail_mem;
  uint64_t* avail_core_lo;
  uint64_t* avail_core_hi;
  uint64_t* avail_gres;
} cns_step_result_soa;

/* One pass of SchedulePendingSteps over all the jobs.  Needs cns_set_nodes (GRES layout).  *kernel_ms (may be NULL):
 * HIP-event time of the device work. */
int cns_schedule_steps(cns_handle* h, const cns_step_job_soa* jobs, const cns_step_soa* steps, cns_step_result_soa* out,
                       double* kernel_ms);

#ifdef __cplusplus
}
#endif
#endif /* CRANE_GPU_STEPS_H_ */
