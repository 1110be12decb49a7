// gfx950 (MI355X / CDNA4) kernels of the node-selection engine.  Hand-written HIP; no MFMA — this is
// integer / bitmask work bound by HBM/L2 latency and LDS, not a dense contraction.
//
// k_init_nodes : one thread per partition slot — NodeSelect's prologue on device:
//                res_avail = res_total - running allocations, the per-node time map and the initial
//                fp64 cost (src/CraneCtld/JobScheduler.cpp:6681-6732, JobScheduler.h:301-338,498-511).
// k_select<NPL>: ONE 1024-thread workgroup per partition (independent LocalScheduler,
//                JobScheduler.cpp:6723-6727), persistent over that partition's whole job queue, because
//                job j+1 depends on job j's commit (SURVEY.md §7 "sequential semantics").  Inside a job
//                the reference's cost-ordered node walk (JobScheduler.cpp:6188-6300) becomes
//                  - a register-resident node tile: each lane owns NPL nodes (cost + "front" summary),
//                  - a per-lane feasibility filter + (cost, index) argmin, wave64 shuffle reduce,
//                    16-entry LDS cross-wave reduce  -> the next node in (cost, idx) order that can pass,
//                  - exact verification of that node by wave 0: wave-parallel window-min (Ckmin) over
//                    the node's time map in HBM, GetFeasibleResourceInNode on bit masks,
//                  - commit by wave 0: wave-parallel sorted-array update of the time map, fp64 cost
//                    update, LDS broadcast of the new summary to the owning lane.
//                Failing that, the first-k-by-total-capacity nodes are selected the same way and the
//                earliest common start is found on their time maps (backfill, JobScheduler.h:792-865).
// Compile with -ffp-contract=off (fp64 cost must match the CPU bit for bit).
#include <hip/hip_runtime.h>

#include "engine_params.h"

namespace cns {

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 cost_key(double c) { return (u64)__double_as_longlong(c); }

__device__ __forceinline__ void wave_argmin(u64& c, u32& p) {
#pragma unroll
  for (int off = 32; off; off >>= 1) {
    u64 oc = __shfl_xor(c, off);
    u32 op = __shfl_xor(p, off);
    bool take = (oc < c) || (oc == c && op < p);
    c = take ? oc : c;
    p = take ? op : p;
  }
}
__device__ __forceinline__ i64 wave_min_i64(i64 v) {
#pragma unroll
  for (int off = 32; off; off >>= 1) { i64 o = __shfl_xor(v, off); v = o < v ? o : v; }
  return v;
}
__device__ __forceinline__ i64 wave_max_i64(i64 v) {
#pragma unroll
  for (int off = 32; off; off >>= 1) { i64 o = __shfl_xor(v, off); v = o > v ? o : v; }
  return v;
}
__device__ __forceinline__ u64 wave_min_u64(u64 v) {
#pragma unroll
  for (int off = 32; off; off >>= 1) { u64 o = __shfl_xor(v, off); v = o < v ? o : v; }
  return v;
}
__device__ __forceinline__ u64 wave_and_u64(u64 v) {
#pragma unroll
  for (int off = 32; off; off >>= 1) v &= __shfl_xor(v, off);
  return v;
}

__device__ __forceinline__ int clamp_cpu(i64 c) {
  return c > 0x7FFFFFFFll ? 0x7FFFFFFF : (c < 0 ? 0 : (int)c);
}
__device__ __forceinline__ u32 mem_mib_ceil(u64 m) {
  u64 v = (m + 0xFFFFFull) >> 20;
  if (m > 0xFFFFFFFFFFF00000ull) v = 0xFFFFFFFFull;
  return v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)v;
}
__device__ __forceinline__ u64 class_counts(u64 gres, const GresDev& L) {
  u64 c = 0;
  for (int g = 0; g < (int)L.num_classes; ++g) c |= (u64)popc64(gres & L.class_mask[g]) << (8 * g);
  return c;
}
__device__ __forceinline__ u32 sum_bytes(u64 v) {
  return __builtin_amdgcn_sad_u8((u32)v, 0u, __builtin_amdgcn_sad_u8((u32)(v >> 32), 0u, 0u));
}
__device__ __forceinline__ void set_fault(const KParams& P, u32 code, u32 a, u32 b, u32 c) {
  if (atomicCAS(P.fault, 0u, code) == 0u) { P.fault[1] = a; P.fault[2] = b; P.fault[3] = c; }
}
__device__ __forceinline__ Res res_zero() { Res r; r.cpu = 0; r.mem = 0; r.clo = 0; r.chi = 0; r.gres = 0; return r; }

// ---------------------------------------------------------------------------------------------
// k_init_nodes — prologue
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_init_nodes(const KParams P) {
  u32 q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= P.num_slots) return;
  const u32 n = P.slot_node[q];
  const Res tot = P.total[n];
  Res a0 = tot;
  double cost = 0.0;
  TlEntry* T = P.tl + (u64)n * P.tl_cap;
  u32 len = 1;  // T[0] reserved for {now, avail0}
  const double tcpu = (double)tot.cpu / 256.0;
  for (u32 a = P.rn_off[n]; a < P.rn_off[n + 1]; ++a) {
    i64 end = P.rn_end[a];
    if (end < P.now + 1) end = P.now + 1;  // JobScheduler.cpp:6513-6514
    const Res r = P.rn_res[a];
    res_sub(a0, r);                        // JobScheduler.h:313
    // NodeRater ctor, JobScheduler.h:508-510 + MinCpuTimeRatioFirst :47-53 (ratio first, then x secs)
    double ratio = ((double)r.cpu / 256.0) / tcpu;
    cost += (double)(end - P.now) * ratio;
    // sorted insert of the release {end, r}; equal end times accumulate (JobScheduler.h:325-335)
    u32 i = 1;
    while (i < len && T[i].t < end) ++i;
    if (i < len && T[i].t == end) {
      res_add(T[i].r, r);
    } else {
      for (u32 m = len; m > i; --m) T[m] = T[m - 1];
      T[i].t = end;
      T[i].r = r;
      ++len;
    }
  }
  T[0].t = P.now;
  T[0].r = a0;
  for (u32 i = 1; i < len; ++i) {  // value at a change time = previous value + what is released there
    Res v = T[i - 1].r;
    res_add(v, T[i].r);
    T[i].r = v;
  }
  T[len].t = kInf;  // time_avail_res_map[end].SetToZero(), JobScheduler.h:337
  T[len].r = res_zero();
  ++len;
  P.tl_len[n] = len;
  P.avail0[n] = a0;
  P.cost[q] = cost;
  P.f_cpu[q] = clamp_cpu(a0.cpu);
  P.f_mem[q] = mem_mib_ceil(a0.mem);
  P.f_cnt[q] = class_counts(a0.gres, P.gres);
}

// ---------------------------------------------------------------------------------------------
// per-job uniform state
// ---------------------------------------------------------------------------------------------
struct JobCtx {
  u64 ji;        // index in the grouped job table
  i64 L, E;      // time_limit, now + time_limit
  Req node_view; // req_node_res_view
  i64 tcpu;      // req_task_res_view
  u64 tmem;
  Req min_view;  // req_node + req_task * tpn_min (JobScheduler.cpp:6154-6156)
  u32 k, ntasks, tmin, tmax, flags;
  bool general;  // ntasks != node_num: capacities matter, priority_queue emulation needed
  u64 incl_b, incl_e, excl_b, excl_e;
};

struct UpdRec {  // what the owning lane must refresh after a commit
  u32 p, len;
  double cost;
  int fcpu;
  u32 fmem;
  u64 fcnt;
  u32 has_front;
  u32 pad;
};

// membership of node n in the job's included / excluded list (JobScheduler.cpp:6202-6220)
__device__ __forceinline__ bool in_list(const u32* lst, u64 b, u64 e, u32 n) {
  for (u64 i = b; i < e; ++i)
    if (lst[i] == n) return true;
  return false;
}

// ---------------------------------------------------------------------------------------------
// wave-0 routines (all 64 lanes of wave 0 execute them; values named "uniform" are wave-uniform)
// ---------------------------------------------------------------------------------------------

// Window-min of node n over [now, E): min_res_on_node = res_avail; for entries with time < E: Ckmin
// (JobScheduler.cpp:6278-6283).  Lanes fold entries in parallel, then a wave64 butterfly.
__device__ __forceinline__ Res window_min(const KParams& P, u32 n, const Res& a0, i64 E, u32 lane) {
  const TlEntry* T = P.tl + (u64)n * P.tl_cap;
  const u32 len = P.tl_len[n];
  i64 cpu = a0.cpu;
  u64 mem = a0.mem, clo = ~0ull, chi = ~0ull, g = a0.gres;
  for (u32 base = 0; base < len; base += 64) {
    u32 i = base + lane;
    bool act = i < len;
    TlEntry e;
    e.t = kInf;
    if (act) e = T[i];
    bool inw = act && e.t < E;
    if (inw) {
      cpu = e.r.cpu < cpu ? e.r.cpu : cpu;
      mem = e.r.mem < mem ? e.r.mem : mem;
      if ((e.r.clo | e.r.chi) != 0) { clo &= e.r.clo; chi &= e.r.chi; }  // empty core set = skipped
      g &= e.r.gres;
    }
    if (__any(act && !inw)) break;  // sorted by time: nothing later is inside the window
  }
  Res m;
  m.cpu = wave_min_i64(cpu);
  m.mem = wave_min_u64(mem);
  clo = wave_and_u64(clo);
  chi = wave_and_u64(chi);
  m.gres = wave_and_u64(g);
  if ((a0.clo | a0.chi) != 0) { m.clo = a0.clo & clo; m.chi = a0.chi & chi; }
  else { m.clo = 0; m.chi = 0; }
  return m;
}

// Exclusive job: every entry with time < E must still hold the whole node (JobScheduler.cpp:6249-6257).
__device__ __forceinline__ bool window_all_total(const KParams& P, u32 n, const Res& tot, i64 E, u32 lane) {
  const TlEntry* T = P.tl + (u64)n * P.tl_cap;
  const u32 len = P.tl_len[n];
  bool bad = false;
  for (u32 base = 0; base < len; base += 64) {
    u32 i = base + lane;
    bool act = i < len;
    TlEntry e;
    e.t = kInf;
    if (act) e = T[i];
    bool inw = act && e.t < E;
    if (inw && !res_le(tot, e.r)) bad = true;
    if (__any(act && !inw)) break;
  }
  return !__any(bad);
}

// NodeState::UpdateResourceInNode (allocate), JobScheduler.h:340-459, on the sorted-array time map.
// Entries with start <= t < end lose `res`; boundaries at start / end are inserted when missing (the
// end boundary copies the un-subtracted value of the entry covering `end`).  Returns the new length.
__device__ u32 tl_commit(const KParams& P, u32 n, i64 start, i64 end, const Res& res, u32 lane, u32 job) {
  TlEntry* T = P.tl + (u64)n * P.tl_cap;
  const u32 len = P.tl_len[n];
  u32 c_start = 0, c_end = 0;  // #entries with t <= start / t <= end
  for (u32 base = 0; base < len; base += 64) {
    u32 i = base + lane;
    bool act = i < len;
    i64 t = act ? T[i].t : kInf;
    c_start += __popcll(__ballot(act && t <= start));
    c_end += __popcll(__ballot(act && t <= end));
    if (__any(act && t > end)) break;
  }
  if (c_start == 0 || c_end >= len || len + 2 > P.tl_cap) {  // cases #1/#2 cannot occur: the INF sentinel is last
    if (lane == 0) set_fault(P, 1, job, n, len);
    return len;
  }
  const u32 ib = c_start - 1, ie0 = c_end - 1;
  const TlEntry eb = T[ib], ee = T[ie0];
  const bool ins_s = eb.t != start, ins_e = ee.t != end;
  const int first_base = (int)(ib / 64) * 64;
  for (int base = (int)((len - 1) / 64) * 64; base >= first_base; base -= 64) {
    u32 i = (u32)base + lane;
    bool act = i < len && i >= ib;
    TlEntry e;
    e.t = kInf;
    e.r = res_zero();
    if (act) e = T[i];
    u32 np = i + ((ins_s && i > ib) ? 1u : 0u) + ((ins_e && i > ie0) ? 1u : 0u);
    bool sub = act && e.t >= start && e.t < end;
    if (sub) res_sub(e.r, res);
    if (act && (np != i || sub)) T[np] = e;
  }
  if (lane == 0) {
    if (ins_s) {
      TlEntry s = eb;
      s.t = start;
      res_sub(s.r, res);
      T[ib + 1] = s;
    }
    if (ins_e) {
      TlEntry x = ee;
      x.t = end;
      T[ie0 + (ins_s ? 1u : 0u) + 1] = x;
    }
    P.tl_len[n] = len + (ins_s ? 1u : 0u) + (ins_e ? 1u : 0u);
  }
  __threadfence_block();
  return len + (ins_s ? 1u : 0u) + (ins_e ? 1u : 0u);
}

// Earliest s >= t such that `alloc` fits node n throughout [s, s + L); kInf if never.
// (per-node half of EarliestStartSubsetSelector, JobScheduler.h:737-790,812-855)
__device__ i64 next_fit(const TlEntry* T, u32 len, const Res& alloc, i64 L, i64 t, u32& j) {
  while (j + 1 < len && T[j + 1].t <= t) ++j;
  i64 s = t;
  u32 i = j;
  while (true) {
    const Res r = T[i].r;
    if (!res_le(alloc, r)) {
      if (i + 1 >= len) return kInf;
      ++i;
      s = T[i].t;
      j = i;
      continue;
    }
    if (i + 1 >= len) return s;  // satisfied through the last entry: iterator "ReachEnd"
    i64 nt = T[i + 1].t;
    if (nt - s >= L) return s;   // kth_time + time_limit <= next flip
    ++i;
  }
}

// Shared by the "start now" and "backfill" endings: H[0..k) holds the selected nodes with their
// assigned task counts; computes the allocations, commits them into the time maps and costs, emits
// the placement records (sorted by node index) and the owner updates.
__device__ void commit_selection(const KParams& P, const JobCtx& J, HeapEnt* H, u32 qbeg, i64 start,
                                 u32 lane, UpdRec* s_upd, int* s_nupd) {
  const i64 end = start + J.L;  // job->end_time = start_time + time_limit, JobScheduler.cpp:6772
  const u32 orig = P.j_orig[J.ji];
  const u64 poff = P.j_place_off[J.ji];
  for (u32 i = 0; i < J.k; ++i) {
    HeapEnt ent = H[i];
    const u32 n = ent.node;
    const Res tot = P.total[n];
    const Res e0 = P.tl[(u64)n * P.tl_cap].r;
    u32 newlen = tl_commit(P, n, start, end, ent.res, lane, orig);
    // MinCpuTimeRatioFirst::UpdateCost, JobScheduler.h:47-53 — ratio first, then x seconds, then +=
    double ratio = ((double)ent.res.cpu / 256.0) / ((double)tot.cpu / 256.0);
    double delta = (double)(end - start) * ratio;
    double ncost = ent.cost + delta;
    UpdRec u;
    u.p = ent.p;
    u.len = newlen;
    u.cost = ncost;
    u.has_front = (start == P.now) ? 1u : 0u;
    Res f = e0;
    if (u.has_front) res_sub(f, ent.res);
    u.fcpu = clamp_cpu(f.cpu);
    u.fmem = mem_mib_ceil(f.mem);
    u.fcnt = class_counts(f.gres, P.gres);
    u.pad = 0;
    if (lane == 0) {
      const u32 q = qbeg + (ent.p >> 10) * 960u + (ent.p & 1023u);
      P.cost[q] = ncost;
      if (u.has_front) { P.f_cpu[q] = u.fcpu; P.f_mem[q] = u.fmem; P.f_cnt[q] = u.fcnt; }
      if (J.k <= (u32)kMaxUpd) s_upd[i] = u;
    }
  }
  if (lane == 0) *s_nupd = J.k <= (u32)kMaxUpd ? (int)J.k : -1;
  // placement records, ascending node index: rank = #selected nodes with a smaller index
  for (u32 i = lane; i < J.k; i += 64) {
    const HeapEnt me = H[i];
    u32 rank = 0;
    for (u32 m = 0; m < J.k; ++m) rank += H[m].node < me.node ? 1u : 0u;
    const u64 o = poff + rank;
    P.o_node[o] = me.node;
    P.o_ntasks[o] = (u32)me.ntasks;
    P.o_cpu[o] = me.res.cpu;
    P.o_mem[o] = me.res.mem;
    P.o_clo[o] = me.res.clo;
    P.o_chi[o] = me.res.chi;
    P.o_gres[o] = me.res.gres;
  }
}

// Task distribution over the k selected nodes, smallest capacity first (JobScheduler.cpp:6304-6325 /
// :6345-6367), then the per-node allocation cut out of ent.res.  Leaves H[i].ntasks = tasks on the node
// and H[i].res = allocated_res on the node.  Returns false on an invariant violation.
__device__ bool distribute_and_alloc(const KParams& P, const JobCtx& J, HeapEnt* H, u32 lane) {
  if (J.general) {
    if (lane == 0) {
      int rest = (int)J.ntasks - (int)J.k;
      for (int len = (int)J.k; len >= 1; --len) {
        int cap = H[0].ntasks;  // top = smallest ntasks_on_node
        int t = (rest < cap - 1 ? rest : cap - 1) + 1;
        rest -= t - 1;
        pq_pop(H, len);  // removed top now sits in H[len-1]
        H[len - 1].ntasks = t;
      }
    }
    __threadfence_block();
  }
  bool ok = true;
  for (u32 i = lane; i < J.k; i += 64) {
    HeapEnt e = H[i];
    if (!J.general) e.ntasks = 1;  // rest_ntasks == 0: min(0, cap-1)+1
    if (!(J.flags & kJfExclusive)) {
      Res a;
      Req v = compose(J.node_view, J.tcpu, J.tmem, (u32)e.ntasks);
      if (!feasible(v, e.res, a, P.gres)) ok = false;  // CRANE_ASSERT_MSG(ok, ...) :6316
      else e.res = a;
    }
    H[i] = e;
  }
  __threadfence_block();
  return !__any(!ok);
}

// ---------------------------------------------------------------------------------------------
// k_select — one persistent workgroup per partition, wave-specialised:
//   wave 0        "worker"  : exact node test, priority_queue emulation, commit, backfill (serial part)
//   waves 1..15   "scanners": hold the partition's node tile in registers (slot p = r*960 + t) and, per
//                             round, deliver the next node in (cost, index) order that passes the filter
// Both roles run the same barrier schedule; they exchange only the per-wave argmin slots, one flag and
// the owner-update records through LDS.  Separate branches => separate register allocation: the tile
// registers are not live in the worker's code and vice versa.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ JobCtx load_job(const KParams& P, u64 ji) {
  JobCtx J;
  J.ji = ji;
  J.L = P.j_L[ji];
  J.E = P.now + J.L;
  J.node_view.cpu = P.j_ncpu[ji];
  J.node_view.mem = P.j_nmem[ji];
  J.node_view.gtot = P.j_gtot[ji];
  J.node_view.gspec = P.j_gspec[ji];
  J.tcpu = P.j_tcpu[ji];
  J.tmem = P.j_tmem[ji];
  J.k = P.j_k[ji];
  J.ntasks = P.j_ntasks[ji];
  J.tmin = P.j_tmin[ji];
  J.tmax = P.j_tmax[ji];
  J.flags = P.j_flags[ji];
  J.general = J.ntasks != J.k;
  J.min_view = compose(J.node_view, J.tcpu, J.tmem, J.tmin);
  J.incl_b = J.incl_e = J.excl_b = J.excl_e = 0;
  if (J.flags & kJfIncl) { J.incl_b = P.j_incl_off[ji]; J.incl_e = P.j_incl_off[ji + 1]; }
  if (J.flags & kJfExcl) { J.excl_b = P.j_excl_off[ji]; J.excl_e = P.j_excl_off[ji + 1]; }
  return J;
}

// requests no node can ever satisfy under the engine's 32-bit front summaries (cpu totals are
// validated < 2^31-1 by cns_set_nodes; a class holds <= 64 slots)
__device__ __forceinline__ bool job_impossible(const JobCtx& J) {
  return J.min_view.cpu > 0x7FFFFFFEll || (J.node_view.gspec & 0x8080808080808080ull) != 0;
}

// ntasks_on_node_total per node type (JobScheduler.cpp:6222): lane t evaluates type t
__device__ __forceinline__ int type_capacity(const KParams& P, const JobCtx& J, const Res& ttot, u32 lane) {
  int tt = 0;
  if (lane < P.num_types) {
    if (J.general) tt = max_tasks(J.min_view, J.tcpu, J.tmem, J.tmin, J.tmax, ttot, P.gres);
    else { Res tmp; tt = feasible(J.min_view, ttot, tmp, P.gres) ? (int)J.tmin : 0; }
  }
  return tt;
}

__device__ __forceinline__ void reduce16(u64& wc, u32& wp) {
#pragma unroll
  for (int off = kWaves / 2; off; off >>= 1) {
    u64 oc = __shfl_xor(wc, off);
    u32 op = __shfl_xor(wp, off);
    bool take = (oc < wc) || (oc == wc && op < wp);
    wc = take ? oc : wc;
    wp = take ? op : wp;
  }
}

constexpr u32 kScan = (kWaves - 1) * 64;  // 960 scanner lanes
__device__ __forceinline__ u32 slot_of_code(u32 code) { return (code >> 10) * kScan + (code & 1023u); }

template <int NPL>
__global__ __launch_bounds__(kBlock) void k_select(const KParams P) {
  const u32 part = blockIdx.x;
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const u32 qbeg = P.part_off[part];
  const u32 nn = P.part_off[part + 1] - qbeg;
  const u64 jbeg = P.pj_off[part], jend = P.pj_off[part + 1];
  if (jbeg >= jend) return;

  __shared__ u64 s_wc[2][kWaves];
  __shared__ u32 s_wp[2][kWaves];
  __shared__ int s_flag;
  __shared__ int s_nupd;
  __shared__ UpdRec s_upd[kMaxUpd];
  __shared__ HeapEnt s_heap[kLdsHeap];

  const Res ttot = lane < P.num_types ? P.type_total[lane] : res_zero();  // lane t holds node type t
  int par = 0;

  if (wave == 0) {
    // =============================================================================================
    // WORKER
    // =============================================================================================
    if (lane == 0) { s_wc[0][0] = ~0ull; s_wc[1][0] = ~0ull; s_wp[0][0] = kNone; s_wp[1][0] = kNone; }
    HeapEnt* const gheap = P.heap + qbeg + part;
    for (u64 ji = jbeg; ji < jend; ++ji) {
      const JobCtx J = load_job(P, ji);
      const bool excl_job = (J.flags & kJfExclusive) != 0;
      const bool impossible = job_impossible(J);
      const u32 orig = P.j_orig[ji];
      const int tt_lane = type_capacity(P, J, ttot, lane);
      HeapEnt* const H = (J.k < (u32)kLdsHeap) ? s_heap : gheap;

      // ---- Phase A: start now (GetNodesAndTrySchedule_, JobScheduler.cpp:6188-6333) -------------
      bool success = false;
      int hsize = 0, hsum = 0;  // topk_nodes_avail.size(), topk_ntasks_sum_avail
      while (!impossible) {
        __syncthreads();  // B1: scanners published their per-wave argmin
        u64 wc = s_wc[par][lane & (kWaves - 1)];
        u32 wcode = s_wp[par][lane & (kWaves - 1)];
        reduce16(wc, wcode);
        par ^= 1;
        if (wcode == kNone) break;  // no node can start this job now
        const u32 wp = slot_of_code(wcode);
        const u32 n = P.slot_node[qbeg + wp];
        int code = 0;
        bool ok = false;
        Res m = res_zero();
        int ta = 0;
        if (!excl_job) {
          const Res a0 = P.avail0[n];
          Res f;
          if (feasible(J.min_view, a0, f, P.gres)) {             // :6274
            m = window_min(P, n, a0, J.E, lane);                 // :6278-6283
            if (J.general) ta = max_tasks(J.min_view, J.tcpu, J.tmem, J.tmin, J.tmax, m, P.gres);  // :6285
            else ta = feasible(J.min_view, m, f, P.gres) ? (int)J.tmin : 0;
            ok = ta > 0;
          }
        } else {
          m = P.total[n];
          ok = window_all_total(P, n, m, J.E, lane);            // :6250-6260
          ta = __shfl(tt_lane, (int)P.ntype[n]);
        }
        if (ok) {
          HeapEnt e;
          e.ntasks = ta; e.p = wcode; e.node = n; e.pad = 0;
          e.cost = __longlong_as_double((long long)wc);
          e.res = m;
          bool done;
          if (!J.general) {
            if (lane == 0) H[hsize] = e;
            ++hsize;
            done = hsize == (int)J.k;   // k nodes with >= 1 task each and ntasks == k: break (:6294-6297)
          } else {
            int nsum = hsum + ta, nsize = hsize + 1;
            if (lane == 0) {
              H[hsize] = e;
              pq_push(H, nsize);                                                // :6288-6289
              if (nsize > (int)J.k) { nsum -= H[0].ntasks; pq_pop(H, nsize); }  // :6290-6293
            }
            nsum = __shfl(nsum, 0);
            if (nsize > (int)J.k) --nsize;
            hsum = nsum; hsize = nsize;
            done = hsize == (int)J.k && (u32)hsum >= J.ntasks;
          }
          __threadfence_block();
          code = 1;
          if (done) {
            if (!distribute_and_alloc(P, J, H, lane)) { if (lane == 0) set_fault(P, 2, orig, n, 0); }
            commit_selection(P, J, H, qbeg, P.now, lane, s_upd, &s_nupd);   // start_time = now (:6326)
            if (lane == 0) { P.o_start[orig] = P.now; P.o_reason[orig] = 0; }
            code = 2;
          }
        }
        if (lane == 0) s_flag = code;
        __syncthreads();  // B2: verdict (and, on success, the owner updates) visible to the scanners
        if (code == 2) { success = true; break; }
      }
      if (success) continue;

      // ---- Phase B: top-k nodes by total capacity, then backfill -----------------------------------
      // (JobScheduler.cpp:6233-6242, :6335-6368, Backfill_ :6371-6376)
      int nsel = 0, tsum = 0;
      bool complete = false;
      while (!impossible) {
        __syncthreads();  // B1
        u64 wc = s_wc[par][lane & (kWaves - 1)];
        u32 wcode = s_wp[par][lane & (kWaves - 1)];
        reduce16(wc, wcode);
        par ^= 1;
        if (wcode == kNone) break;
        const u32 n = P.slot_node[qbeg + slot_of_code(wcode)];
        HeapEnt e;
        e.p = wcode; e.node = n; e.pad = 0;
        e.cost = __longlong_as_double((long long)wc);
        e.res = res_zero();
        if (!J.general) {
          e.ntasks = 1;
          if (lane == 0) H[nsel] = e;
          ++nsel;
          if (nsel == (int)J.k) { complete = true; break; }
        } else {
          const int tt = __shfl(tt_lane, (int)P.ntype[n]);
          e.ntasks = tt;
          int nsum = tsum + tt, nsize = nsel + 1;  // the push condition (:6233-6234) held, else we had stopped
          if (lane == 0) {
            H[nsel] = e;
            pq_push(H, nsize);
            if (nsize > (int)J.k) { nsum -= H[0].ntasks; pq_pop(H, nsize); }
          }
          nsum = __shfl(nsum, 0);
          if (nsize > (int)J.k) --nsize;
          tsum = nsum; nsel = nsize;
          const bool stop = nsel == (int)J.k && (u32)tsum >= J.ntasks;
          if (lane == 0) s_flag = stop ? 1 : 0;
          __syncthreads();  // B2 (general path only)
          if (stop) { complete = true; break; }
        }
      }
      int code = 0;
      if (complete) {
        __threadfence_block();
        for (u32 i = lane; i < J.k; i += 64) { HeapEnt e = H[i]; e.res = P.total[e.node]; H[i] = e; P.bf_j[qbeg + i] = 0; }
        __threadfence_block();
        if (!distribute_and_alloc(P, J, H, lane)) { if (lane == 0) set_fault(P, 3, orig, 0, 0); }
        // EarliestStartSubsetSelector::CalcEarliestStartTime as a fixed point over the k nodes
        i64 t = P.now;
        bool found = false;
        for (u32 iter = 0; iter < (1u << 22); ++iter) {
          i64 T = t;
          for (u32 i = lane; i < J.k; i += 64) {
            const HeapEnt e = H[i];
            u32 j = P.bf_j[qbeg + i];
            i64 s = next_fit(P.tl + (u64)e.node * P.tl_cap, P.tl_len[e.node], e.res, J.L, t, j);
            P.bf_j[qbeg + i] = j;
            T = s > T ? s : T;
          }
          T = wave_max_i64(T);
          if (T == kInf || T - P.now > P.max_window) break;  // kAlgoMaxTimeWindow, JobScheduler.h:815
          if (T == t) { found = true; break; }
          t = T;
        }
        if (found) {
          int reason = 0;
          if (t != P.now) {  // JobScheduler.cpp:6797-6833 (no reservations in this slice)
            bool notle = false;
            for (u32 i = lane; i < J.k; i += 64) {
              const HeapEnt e = H[i];
              if (!res_le(e.res, P.avail0[e.node])) notle = true;
            }
            reason = __any(notle) ? 2 /*Resource*/ : 1 /*Priority*/;
          }
          commit_selection(P, J, H, qbeg, t, lane, s_upd, &s_nupd);
          if (lane == 0) { P.o_start[orig] = t; P.o_reason[orig] = (uint8_t)reason; }
          code = 2;
        }
      }
      if (code == 0 && lane == 0) { P.o_start[orig] = 0; P.o_reason[orig] = 2; }  // "Resource", :6768
      if (lane == 0) s_flag = code;
      __syncthreads();  // B3
    }
  } else {
    // =============================================================================================
    // SCANNERS — register-resident node tile: slot p = r*960 + t, code = r<<10 | t
    // =============================================================================================
    const u32 t = tid - 64u;
    double cost[NPL];
    int fcpu[NPL];
    u32 fmem[NPL];
    u64 fcnt[NPL];
    u32 meta[NPL];  // bit31 valid | len << 8 | type
#pragma unroll
    for (int r = 0; r < NPL; ++r) {
      const u32 p = (u32)r * kScan + t;
      if (p < nn) {
        const u32 q = qbeg + p;
        const u32 n = P.slot_node[q];
        cost[r] = P.cost[q];
        fcpu[r] = P.f_cpu[q];
        fmem[r] = P.f_mem[q];
        fcnt[r] = P.f_cnt[q];
        meta[r] = 0x80000000u | (P.tl_len[n] << 8) | (u32)P.ntype[n];
      } else {
        cost[r] = 0.0; fcpu[r] = 0; fmem[r] = 0; fcnt[r] = 0; meta[r] = 0;
      }
    }
    const int ttot_cpu32 = clamp_cpu(ttot.cpu);
    const u32 ttot_mem32 = mem_mib_ceil(ttot.mem);
    const u64 ttot_cnt = class_counts(ttot.gres, P.gres);
    const u64 H8 = 0x8080808080808080ull;

    for (u64 ji = jbeg; ji < jend; ++ji) {
      const JobCtx J = load_job(P, ji);
      const bool excl_job = (J.flags & kJfExclusive) != 0;
      const bool has_lists = (J.flags & (kJfIncl | kJfExcl)) != 0;
      const bool has_gres = (J.flags & kJfGres) != 0;
      const bool impossible = job_impossible(J);
      const u64 typeok = __ballot(type_capacity(P, J, ttot, lane) > 0);
      const int req_cpu32 = J.min_view.cpu > 0x7FFFFFFFll ? 0x7FFFFFFF : (int)J.min_view.cpu;
      const u32 req_mem32 = (J.min_view.mem >> 20) > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)(J.min_view.mem >> 20);

      u32 used = 0;
      int verdict = 0;
      // ---- Phase A ----------------------------------------------------------------------------------
      while (!impossible) {
        u64 bc = ~0ull;
        u32 bp = kNone;
#pragma unroll
        for (int r = 0; r < NPL; ++r) {
          const u32 m = meta[r];
          const u32 ty = m & 0xFFu;
          bool c = (m >> 31) && !((used >> r) & 1u) && ((typeok >> ty) & 1ull) &&
                   (((m >> 8) & 0xFFFFu) < P.max_jobs_per_node);  // :6194
          if (!excl_job) {
            // necessary for :6274-6285: the entry at `now` lies inside every window
            c = c && req_cpu32 <= fcpu[r] && req_mem32 <= fmem[r];
            if (has_gres) {
              const u64 cn = fcnt[r];
              c = c && ((((cn | H8) - J.node_view.gspec) & H8) == H8);
#pragma unroll
              for (int a = 0; a < kMaxNames; ++a) {
                const u32 tot = (J.node_view.gtot >> (8 * a)) & 0xFFu;
                if (tot) c = c && sum_bytes(cn & P.gres.name_bytes[a]) >= tot;
              }
            }
          } else {  // exclusive: the node must be completely free now (necessary for :6251-6257)
            const int tc = __shfl(ttot_cpu32, (int)ty);
            const u32 tm = __shfl(ttot_mem32, (int)ty);
            const u64 tn = __shfl(ttot_cnt, (int)ty);
            c = c && fcpu[r] >= tc && fmem[r] >= tm && fcnt[r] == tn;
          }
          if (c && has_lists) {
            const u32 n = P.slot_node[qbeg + (u32)r * kScan + t];
            if ((J.flags & kJfIncl) && !in_list(P.incl_nodes, J.incl_b, J.incl_e, n)) c = false;
            if ((J.flags & kJfExcl) && in_list(P.excl_nodes, J.excl_b, J.excl_e, n)) c = false;
          }
          const u64 ck = cost_key(cost[r]);
          const u32 code = ((u32)r << 10) | t;
          if (c && (ck < bc || (ck == bc && code < bp))) { bc = ck; bp = code; }
        }
        wave_argmin(bc, bp);
        if (lane == 0) { s_wc[par][wave] = bc; s_wp[par][wave] = bp; }
        __syncthreads();  // B1
        u64 wc = s_wc[par][lane & (kWaves - 1)];
        u32 wcode = s_wp[par][lane & (kWaves - 1)];
        reduce16(wc, wcode);
        par ^= 1;
        if (wcode == kNone) break;
        if ((wcode & 1023u) == t) used |= 1u << (wcode >> 10);
        __syncthreads();  // B2
        verdict = s_flag;
        if (verdict == 2) break;
      }
      // ---- Phase B ----------------------------------------------------------------------------------
      if (verdict != 2) {
        used = 0;
        int nsel = 0;
        while (!impossible) {
          u64 bc = ~0ull;
          u32 bp = kNone;
#pragma unroll
          for (int r = 0; r < NPL; ++r) {
            const u32 m = meta[r];
            bool c = (m >> 31) && !((used >> r) & 1u) && ((typeok >> (m & 0xFFu)) & 1ull) &&
                     (((m >> 8) & 0xFFFFu) < P.max_jobs_per_node);
            if (c && has_lists) {
              const u32 n = P.slot_node[qbeg + (u32)r * kScan + t];
              if ((J.flags & kJfIncl) && !in_list(P.incl_nodes, J.incl_b, J.incl_e, n)) c = false;
              if ((J.flags & kJfExcl) && in_list(P.excl_nodes, J.excl_b, J.excl_e, n)) c = false;
            }
            const u64 ck = cost_key(cost[r]);
            const u32 code = ((u32)r << 10) | t;
            if (c && (ck < bc || (ck == bc && code < bp))) { bc = ck; bp = code; }
          }
          wave_argmin(bc, bp);
          if (lane == 0) { s_wc[par][wave] = bc; s_wp[par][wave] = bp; }
          __syncthreads();  // B1
          u64 wc = s_wc[par][lane & (kWaves - 1)];
          u32 wcode = s_wp[par][lane & (kWaves - 1)];
          reduce16(wc, wcode);
          par ^= 1;
          if (wcode == kNone) break;
          if ((wcode & 1023u) == t) used |= 1u << (wcode >> 10);
          if (!J.general) {
            if (++nsel == (int)J.k) break;
          } else {
            __syncthreads();  // B2 (general path only)
            if (s_flag == 1) break;
          }
        }
        __syncthreads();  // B3: worker finished backfill + commit (or gave up)
        verdict = s_flag;
      }
      // ---- owners refresh their registers ---------------------------------------------------------------
      if (verdict == 2) {
        const int nu = s_nupd;
        if (nu >= 0) {
          for (int i = 0; i < nu; ++i) {
            const UpdRec u = s_upd[i];
            if ((u.p & 1023u) == t) {
              const int rr = (int)(u.p >> 10);
#pragma unroll
              for (int r = 0; r < NPL; ++r)
                if (r == rr) {
                  cost[r] = u.cost;
                  meta[r] = (meta[r] & 0x800000FFu) | (u.len << 8);
                  if (u.has_front) { fcpu[r] = u.fcpu; fmem[r] = u.fmem; fcnt[r] = u.fcnt; }
                }
            }
          }
        } else {  // more nodes than the LDS broadcast holds: reload the tile from HBM
#pragma unroll
          for (int r = 0; r < NPL; ++r) {
            const u32 p = (u32)r * kScan + t;
            if (p < nn) {
              const u32 q = qbeg + p;
              cost[r] = P.cost[q]; fcpu[r] = P.f_cpu[q]; fmem[r] = P.f_mem[q]; fcnt[r] = P.f_cnt[q];
              meta[r] = (meta[r] & 0x800000FFu) | (P.tl_len[P.slot_node[q]] << 8);
            }
          }
        }
      }
    }
  }
}

template __global__ void k_select<1>(const KParams);
template __global__ void k_select<2>(const KParams);
template __global__ void k_select<3>(const KParams);
template __global__ void k_select<5>(const KParams);
template __global__ void k_select<9>(const KParams);
template __global__ void k_select<18>(const KParams);

}  // namespace cns
