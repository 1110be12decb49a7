// Step scheduler on the GPU (include/crane_gpu/steps.h): JobInCtld::SchedulePendingSteps
// (src/CraneCtld/CtldPublicDefs.cpp:2038-2159) for every job with pending steps.
//
// A job's step queue is a chain (a step takes resources out of step_res_avail_, the next one sees what is left, the
// first one that does not fit stops the queue), but jobs never share an allocation: ONE THREAD PER JOB, thousands of
// independent chains.  The work per node is the exact bit-mask algebra of the node-selection path (res_dev.h:
// GetFeasibleResourceInNode, -=), the top-k queue is libstdc++'s heap move for move (as in pq_emul.h, on an 8-byte
// entry).  Availability lives in HBM as one 40-byte Res per (job, node), updated in place.
//
// Included by engine.hip (one translation unit, namespace cns).
#pragma once

namespace cns {

struct StepEnt { u32 ntasks; u32 pos; };   // NodeInfo {ntasks_on_node, craned_id} (:2056-2062)
// a < b  <=>  a.ntasks_on_node > b.ntasks_on_node (:2059-2061)
__device__ __forceinline__ bool step_comp(const StepEnt& a, const StepEnt& b) { return a.ntasks > b.ntasks; }
// std::__push_heap / std::__adjust_heap of GCC's bits/stl_heap.h (see pq_emul.h for the annotated form)
__device__ __forceinline__ void step_push_up(StepEnt* first, int hole, int top, StepEnt value) {
  int parent = (hole - 1) / 2;
  while (hole > top && step_comp(first[parent], value)) {
    first[hole] = first[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  first[hole] = value;
}
__device__ __forceinline__ void step_pq_push(StepEnt* first, int len) { step_push_up(first, len - 1, 0, first[len - 1]); }
__device__ __forceinline__ void step_pq_pop(StepEnt* first, int len) {  // len = size before the pop
  if (len <= 1) return;
  const StepEnt value = first[len - 1];
  first[len - 1] = first[0];
  const int n = len - 1;
  int hole = 0, child = 0;
  while (child < (n - 1) / 2) {
    child = 2 * (child + 1);
    if (step_comp(first[child], first[child - 1])) child--;
    first[hole] = first[child];
    hole = child;
  }
  if ((n & 1) == 0 && child == (n - 2) / 2) {
    child = 2 * (child + 1);
    first[hole] = first[child - 1];
    hole = child - 1;
  }
  step_push_up(first, hole, 0, value);
}

struct StepRec {
  Req node_req, task_req;      // req_node_res_view, req_task_res_view
  u32 node_num, ntasks, tmin, tmax;
  u32 incl_b, incl_e, excl_b, excl_e;
  u64 place_off, task_off;
};

struct StepParams {
  u32 num_jobs, pad;
  const u32* node_off; const u32* node_idx; Res* avail; const u32* step_off;
  const StepRec* steps; const u32* incl; const u32* excl;
  uint8_t* scheduled; u32* o_node; u32* o_nt; Res* o_alloc; u32* t_node; Res* t_alloc;
  GresDev gres;
};

constexpr int kStepMaxNodes = 64;   // CNS_STEP_MAX_NODES

__global__ __launch_bounds__(64) void k_sched_steps(const StepParams P) {
  const u32 j = blockIdx.x * 64 + threadIdx.x;
  if (j >= P.num_jobs) return;
  const u32 nb = P.node_off[j], ne = P.node_off[j + 1];
  StepEnt heap[kStepMaxNodes + 1];
  for (u32 s = P.step_off[j]; s < P.step_off[j + 1]; ++s) {
    const StepRec st = P.steps[s];
    int len = 0;
    u32 sum = 0;
    for (u32 pos = nb; pos < ne; ++pos) {                                     // :2066-2102
      const u32 n = P.node_idx[pos];
      bool skip = false;
      for (u32 x = st.excl_b; x < st.excl_e; ++x) skip |= P.excl[x] == n;
      if (st.incl_e > st.incl_b) {
        bool in = false;
        for (u32 x = st.incl_b; x < st.incl_e; ++x) in |= P.incl[x] == n;
        skip |= !in;
      }
      if (skip) continue;
      Res a = P.avail[pos], f;
      if (!feasible(st.node_req, a, f, P.gres)) continue;
      res_sub(a, f);
      u32 nt = 0;
      while (nt < st.tmax && feasible(st.task_req, a, f, P.gres)) { ++nt; res_sub(a, f); }
      if (nt < st.tmin) continue;
      heap[len].ntasks = nt; heap[len].pos = pos;
      ++len;
      step_pq_push(heap, len);
      sum += nt;
      if (len > (int)st.node_num) { sum -= heap[0].ntasks; step_pq_pop(heap, len); --len; }
      if (len == (int)st.node_num && sum >= st.ntasks) break;
    }
    if (len < (int)st.node_num || sum < st.ntasks) break;                     // :2104-2106: the queue stops here
    u32 rest = st.ntasks - st.node_num;                                       // :2107
    u64 p = st.place_off, t = st.task_off;
    while (len > 0) {                                                         // :2109-2128
      const StepEnt info = heap[0];
      const u32 node = P.node_idx[info.pos];
      Res ra = P.avail[info.pos], f = res_zero(), total = res_zero();
      feasible(st.node_req, ra, f, P.gres);
      res_sub(ra, f);
      res_add(total, f);
      const u32 nton = (rest < info.ntasks - 1 ? rest : info.ntasks - 1) + 1;
      for (u32 i = 0; i < nton; ++i) {
        feasible(st.task_req, ra, f, P.gres);
        res_sub(ra, f);
        P.t_node[t] = node;
        P.t_alloc[t] = f;
        ++t;
        res_add(total, f);
      }
      rest -= nton - 1;
      P.avail[info.pos] = ra;
      P.o_node[p] = node; P.o_nt[p] = nton; P.o_alloc[p] = total;
      ++p;
      step_pq_pop(heap, len);
      --len;
    }
    P.scheduled[s] = 1;
  }
}

}  // namespace cns
