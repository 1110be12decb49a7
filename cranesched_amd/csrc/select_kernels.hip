// gfx950 (MI355X / CDNA4) kernels of the node-selection engine.  Hand-written HIP; no MFMA — this is
// integer / bitmask work bound by the latency of a serial chain and by one CU's VALU, not a dense contraction.
//
// k_init_nodes : one thread per slot (a partition's node, or the virtual node of a reservation) — NodeSelect's
//                prologue on device: res_avail = res_total - active reservations - running allocations, the
//                per-node time map incl. future-reservation dips, the initial fp64 cost and the node's first
//                reservation (src/CraneCtld/JobScheduler.cpp:6619-6732, JobScheduler.h:301-338,498-511).
// k_prep_jobs  : one thread per pending job — everything about a job that does not depend on the evolving
//                node state (minimum view, request side of the scanners' filters, GRES request shape, the
//                "fits res_total" mask over the node types), written into the job record.
// k_select<NPL>: ONE 512-thread workgroup per partition (independent LocalScheduler, JobScheduler.cpp:6723-6732),
//                persistent over that partition's whole job queue, because job j+1 depends on job j's commit
//                (SURVEY.md §7 "sequential semantics").  8 waves = 2 per SIMD = 256 VGPRs per lane.  The
//                workgroup is wave-specialised:
//                  waves 1..7 "scanners": the partition's node tile lives in their registers (each lane owns
//                     NPL nodes: fp64 cost + a "front" summary of what is free now).  They filter their nodes
//                     against a job and deliver — wave64 DPP argmin, then an LDS exchange — the next node in
//                     (cost, index) order that may pass, i.e. the reference's cost-ordered walk
//                     (JobScheduler.cpp:6188-6300) without the walk.  While the worker handles job j they
//                     PRE-SCAN job j+1 over every node job j cannot change.
//                  wave 0 "worker": exact test of the delivered node — ONE coalesced read of its NodeBlock
//                     (header + time map, lane i <- entry i), wave-parallel window-min (Ckmin),
//                     GetFeasibleResourceInNode on bit masks — then the commit straight from registers:
//                     sorted-array update of the time map, fp64 cost update, LDS broadcast of the new summary
//                     to the owning scanner lane; then it merges the one changed node into the pre-scan of
//                     the next job.  Jobs that cannot start now get their nodes by total capacity from the
//                     same scan and the earliest start by a bit-scan over the time map's "alloc fits" ballot
//                     (backfill, JobScheduler.h:792-865).
//                  multi-node jobs: per-wave top-k lists, k-way merge by the worker, one helper wave per node
//                     for the exact tests and the commits.
// Compile with -ffp-contract=off (fp64 cost must match the CPU bit for bit).
#include <hip/hip_runtime.h>

#include "engine_params.h"


namespace cns {

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 cost_key(double c) { return (u64)__double_as_longlong(c); }
// ... and an order-preserving key for costs of EITHER sign: a cycle with preemption releases cost (UpdateCost(...,
// is_release)), a job preempted twice even below zero, where the raw bit pattern orders the wrong way round.
// smode = ~0: signed mode; 0: the raw pattern (costs only grow from >= 0 in every other cycle).
__device__ __forceinline__ u64 cost_key_m(double c, u64 smode) {
  const u64 b = (u64)__double_as_longlong(c);
  return b ^ (smode & ((u64)((i64)b >> 63) | 0x8000000000000000ull));
}

__device__ __forceinline__ double cost_of_key_m(u64 key, u64 smode) {   // the cost behind such a key
  return __longlong_as_double((long long)(key ^ (smode & ((u64)((i64)~key >> 63) | 0x8000000000000000ull))));
}
// A cycle with preemption (KParams::general_only) sends every job through k_select's general path — except the jobs that can
// preempt nobody (kJfMayPreempt clear: TryPreempt_ returns at once for them, JobScheduler.cpp:6384-6385) and have the shape of
// the inline path: one node, ntasks == node_num, tpn_min == 1, not exclusive.  Worker and scanners evaluate this on the same
// record, so both sides agree per job on the path, the barrier schedule and where the owner updates go (LDS / HBM).
__device__ __forceinline__ bool inline_in_preempt_cycle(u32 flags, u32 k, bool general, bool tmin1) {
  return k == 1 && !general && tmin1 && !(flags & (kJfExclusive | kJfMayPreempt));
}
// ... and node_num 2 .. kMultiK of that shape (the parallel protocol of the multi-node jobs: its commits join the nodes' qos_job_map
// like commit_selection's, pre_join_multi; the owner updates stay in LDS)
__device__ __forceinline__ bool fast_in_preempt_cycle(u32 flags, u32 k, bool general, bool tmin1) {
  return k >= 1 && k <= (u32)kMultiK && !general && tmin1 && !(flags & (kJfExclusive | kJfMayPreempt));
}
// ... and in a group of partitions that share nodes the jobs of that shape keep the inline path too (the commit then lists the node's
// other slots among the owner updates: commit_single_shared); everything else of such a group, and all of it in a cycle with
// preemption, takes the general path.  true = general path ("shared_nodes" in worker and scanners).
__device__ __forceinline__ bool general_path_job(bool general_only, bool shared_group, u32 flags, u32 k, bool general, bool tmin1) {
  if (general_only) return shared_group || !fast_in_preempt_cycle(flags, k, general, tmin1);
  // (node_num 2 .. kMultiK of that shape: the parallel protocol, whose helper waves list the sibling slots as well)
  return shared_group && !(k >= 1 && k <= (u32)kMultiK && !general && tmin1 && !(flags & kJfExclusive));
}

__device__ __forceinline__ u32 uni32(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u64 uni64(u64 v) { return ((u64)uni32((u32)(v >> 32)) << 32) | uni32((u32)v); }
__device__ __forceinline__ u32 rl32(u32 v, u32 idx) { return (u32)__builtin_amdgcn_readlane((int)v, (int)idx); }
__device__ __forceinline__ u64 rl64(u64 v, u32 idx) { return ((u64)rl32((u32)(v >> 32), idx) << 32) | rl32((u32)v, idx); }
__device__ __forceinline__ Res rl_res(const Res& r, u32 idx) {
  Res o;
  o.cpu = (i64)rl64((u64)r.cpu, idx);
  o.mem = rl64(r.mem, idx);
  o.clo = rl64(r.clo, idx);
  o.chi = rl64(r.chi, idx);
  o.gres = rl64(r.gres, idx);
  o.c2 = rl64(r.c2, idx);
  o.c3 = rl64(r.c3, idx);
  return o;
}

// A kernel's by-value KParams must not have its address taken: handed by reference to an out-of-line routine it becomes a copy in
// SCRATCH for the WHOLE kernel — every `P.field`, also in the inlined hot loops, then is a load from private memory, and every
// pointer read from it a generic pointer whose dereference is a flat access with a full `s_waitcnt vmcnt(0) lgkmcnt(0)`.  The
// out-of-line routines therefore take the block's copy in HBM (KParams* Pg, uploaded before the launch; C4 296 -> 276 ms when
// the testers of k_wide followed that rule too).  What they read through it are generic pointers all the same: as_global() says
// where such a pointer points (global_load with counted waits instead of flat_load).
#ifndef CNS_NO_AS_GLOBAL
template <class T>
__device__ __forceinline__ const __attribute__((address_space(1))) T* as_global(const T* p) {
  return (const __attribute__((address_space(1))) T*)p;
}
#else   // (A/B builds)
template <class T> __device__ __forceinline__ const T* as_global(const T* p) { return p; }
#endif

// ... and kparams_scalar() brings the block itself back into SCALAR registers inside such a routine: a copy through the constant
// address space at a readfirstlane'd address, which the compiler splits into s_load's of the fields the routine uses (batched at the
// entry, served by the scalar cache) — instead of one flat_load + `s_waitcnt vmcnt(0) lgkmcnt(0)` per pointer read from *Pg, each of
// which also waited for every store in flight (the testers' commit: ten of them in a row behind the time-map stores; profiles/
// r04_tester_phases.txt).  Rules for the copy: hand it only to __forceinline__ routines (its address must not escape), and index
// nothing in it dynamically — the GRES tables are read through Pg->gres.  (The block is written by the host before the launch
// and by nobody during it.)
#ifndef CNS_NO_KPARAMS_SCALAR
__device__ __forceinline__ KParams kparams_scalar(const KParams* Pg) {
  typedef const __attribute__((address_space(4))) u64* ConstU64;
  static_assert(sizeof(KParams) % 8 == 0, "copied as 64-bit words");
  const ConstU64 s = (ConstU64)(uintptr_t)uni64((u64)(uintptr_t)Pg);
  KParams P;
  u64* const d = (u64*)&P;
#pragma unroll
  for (u32 i = 0; i < (u32)(sizeof(KParams) / 8); ++i) d[i] = s[i];
  return P;
}
#else   // (A/B builds: the routines read the block in HBM field by field, as before)
__device__ __forceinline__ const KParams& kparams_scalar(const KParams* Pg) { return *Pg; }
#endif

// wave-uniform values -> SGPRs, so that the worker's GetFeasibleResourceInNode arithmetic runs on the
// scalar unit and does not compete for vector registers
__device__ __forceinline__ Res uni_res(const Res& r) {
  Res o;
  o.cpu = (i64)uni64((u64)r.cpu); o.mem = uni64(r.mem); o.clo = uni64(r.clo); o.chi = uni64(r.chi); o.gres = uni64(r.gres);
  o.c2 = uni64(r.c2); o.c3 = uni64(r.c3);
  return o;
}

// Workgroup barrier that only drains LDS traffic.  __syncthreads() also waits for every outstanding
// global load / store of the wave (s_waitcnt vmcnt(0)): that would expose, at EVERY barrier, the latency of
// the job-record prefetch (scanners) and of the time-map stores just issued (worker).  Everything the
// roles exchange goes through LDS; the two places where data crosses waves through HBM (owner-update
// lists longer than the LDS list) fence explicitly.
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// The worker re-reads node blocks it stored to in earlier jobs: make those stores complete first.
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- wave64 reductions on the DPP crossbar (no LDS traffic, no ds_bpermute latency) -------------------
// row_shr:1,2,4,8 folds each row of 16 lanes into its lane 15; row_bcast:15 / row_bcast:31 fold the
// four rows into lane 63 (the classic GCN/CDNA wave reduction); v_readlane 63 makes the result scalar.
// Lanes whose DPP source is out of range keep `identity` (bound_ctrl = 0), so op(v, identity) == v.
#define CNS_DPP_STEP32(op, v, ident, ctrl, rmask) \
  v = op(v, (u32)__builtin_amdgcn_update_dpp((int)(ident), (int)(v), ctrl, rmask, 0xF, false))
__device__ __forceinline__ u32 op_umin(u32 a, u32 b) { return a < b ? a : b; }
__device__ __forceinline__ u32 op_umax(u32 a, u32 b) { return a > b ? a : b; }
__device__ __forceinline__ u32 op_and(u32 a, u32 b) { return a & b; }
__device__ __forceinline__ u32 wave_umin32(u32 v) {
  CNS_DPP_STEP32(op_umin, v, 0xFFFFFFFFu, 0x111, 0xF);
  CNS_DPP_STEP32(op_umin, v, 0xFFFFFFFFu, 0x112, 0xF);
  CNS_DPP_STEP32(op_umin, v, 0xFFFFFFFFu, 0x114, 0xF);
  CNS_DPP_STEP32(op_umin, v, 0xFFFFFFFFu, 0x118, 0xF);
  CNS_DPP_STEP32(op_umin, v, 0xFFFFFFFFu, 0x142, 0xA);
  CNS_DPP_STEP32(op_umin, v, 0xFFFFFFFFu, 0x143, 0xC);
  return rl32(v, 63);
}
__device__ __forceinline__ u32 wave_umax32(u32 v) {
  CNS_DPP_STEP32(op_umax, v, 0u, 0x111, 0xF);
  CNS_DPP_STEP32(op_umax, v, 0u, 0x112, 0xF);
  CNS_DPP_STEP32(op_umax, v, 0u, 0x114, 0xF);
  CNS_DPP_STEP32(op_umax, v, 0u, 0x118, 0xF);
  CNS_DPP_STEP32(op_umax, v, 0u, 0x142, 0xA);
  CNS_DPP_STEP32(op_umax, v, 0u, 0x143, 0xC);
  return rl32(v, 63);
}
__device__ __forceinline__ u32 wave_and32(u32 v) {
  CNS_DPP_STEP32(op_and, v, 0xFFFFFFFFu, 0x111, 0xF);
  CNS_DPP_STEP32(op_and, v, 0xFFFFFFFFu, 0x112, 0xF);
  CNS_DPP_STEP32(op_and, v, 0xFFFFFFFFu, 0x114, 0xF);
  CNS_DPP_STEP32(op_and, v, 0xFFFFFFFFu, 0x118, 0xF);
  CNS_DPP_STEP32(op_and, v, 0xFFFFFFFFu, 0x142, 0xA);
  CNS_DPP_STEP32(op_and, v, 0xFFFFFFFFu, 0x143, 0xC);
  return rl32(v, 63);
}
// min over the 16 lanes of each row (used where every row holds the same 16 values)
__device__ __forceinline__ u32 row_umin32(u32 v) {
  CNS_DPP_STEP32(op_umin, v, 0xFFFFFFFFu, 0x111, 0xF);
  CNS_DPP_STEP32(op_umin, v, 0xFFFFFFFFu, 0x112, 0xF);
  CNS_DPP_STEP32(op_umin, v, 0xFFFFFFFFu, 0x114, 0xF);
  CNS_DPP_STEP32(op_umin, v, 0xFFFFFFFFu, 0x118, 0xF);
  return rl32(v, 15);
}
// 64-bit unsigned min / max as cascaded 32-bit reductions: high words first, then the low words of the
// lanes that tie on the high word.  All results are wave-uniform.
__device__ __forceinline__ u64 wave_min_u64(u64 v) {
  const u32 hi = (u32)(v >> 32), lo = (u32)v;
  const u32 mh = wave_umin32(hi);
  const u32 ml = wave_umin32(hi == mh ? lo : 0xFFFFFFFFu);
  return ((u64)mh << 32) | ml;
}
__device__ __forceinline__ u64 wave_max_u64(u64 v) {
  const u32 hi = (u32)(v >> 32), lo = (u32)v;
  const u32 mh = wave_umax32(hi);
  const u32 ml = wave_umax32(hi == mh ? lo : 0u);
  return ((u64)mh << 32) | ml;
}
// order-preserving map i64 -> u64
__device__ __forceinline__ i64 wave_min_i64(i64 v) { return (i64)(wave_min_u64((u64)v ^ 0x8000000000000000ull) ^ 0x8000000000000000ull); }
__device__ __forceinline__ u64 wave_and_u64(u64 v) {
  return ((u64)wave_and32((u32)(v >> 32)) << 32) | wave_and32((u32)v);
}
__device__ __forceinline__ u64 wave_or_u64(u64 v) { return ~wave_and_u64(~v); }
// lexicographic argmin of (cost key, code) over the wave: three cascaded 32-bit minima
__device__ __forceinline__ void wave_argmin(u64& c, u32& p) {
  // (measured: leaving the cascade as soon as one lane is left — ballot, popcount, branch — costs more than the 6 or 12
  // DPP steps it saves: C4 337 vs 328 ms)
  const u32 hi = (u32)(c >> 32), lo = (u32)c;
  const u32 mh = wave_umin32(hi);
  const bool e1 = hi == mh;
  const u32 ml = wave_umin32(e1 ? lo : 0xFFFFFFFFu);
  const bool e2 = e1 & (lo == ml);
  const u32 mp = wave_umin32(e2 ? p : 0xFFFFFFFFu);
  c = ((u64)mh << 32) | ml;
  p = mp;
}
// two independent argmins in one pass: the DPP steps of the two cascades alternate, so that each hides the other's wait states
// (a lone wave per SIMD has nothing else to issue)
__device__ __forceinline__ void wave_umin32x2(u32& a, u32& b) {
  CNS_DPP_STEP32(op_umin, a, 0xFFFFFFFFu, 0x111, 0xF); CNS_DPP_STEP32(op_umin, b, 0xFFFFFFFFu, 0x111, 0xF);
  CNS_DPP_STEP32(op_umin, a, 0xFFFFFFFFu, 0x112, 0xF); CNS_DPP_STEP32(op_umin, b, 0xFFFFFFFFu, 0x112, 0xF);
  CNS_DPP_STEP32(op_umin, a, 0xFFFFFFFFu, 0x114, 0xF); CNS_DPP_STEP32(op_umin, b, 0xFFFFFFFFu, 0x114, 0xF);
  CNS_DPP_STEP32(op_umin, a, 0xFFFFFFFFu, 0x118, 0xF); CNS_DPP_STEP32(op_umin, b, 0xFFFFFFFFu, 0x118, 0xF);
  CNS_DPP_STEP32(op_umin, a, 0xFFFFFFFFu, 0x142, 0xA); CNS_DPP_STEP32(op_umin, b, 0xFFFFFFFFu, 0x142, 0xA);
  CNS_DPP_STEP32(op_umin, a, 0xFFFFFFFFu, 0x143, 0xC); CNS_DPP_STEP32(op_umin, b, 0xFFFFFFFFu, 0x143, 0xC);
  a = rl32(a, 63); b = rl32(b, 63);
}
__device__ __forceinline__ void wave_argmin2(u64& c1, u32& p1, u64& c2, u32& p2) {
  const u32 h1 = (u32)(c1 >> 32), l1 = (u32)c1, h2 = (u32)(c2 >> 32), l2 = (u32)c2;
  u32 mh1 = h1, mh2 = h2;
  wave_umin32x2(mh1, mh2);
  const bool e1 = h1 == mh1, f1 = h2 == mh2;
  u32 ml1 = e1 ? l1 : 0xFFFFFFFFu, ml2 = f1 ? l2 : 0xFFFFFFFFu;
  wave_umin32x2(ml1, ml2);
  const bool e2 = e1 & (l1 == ml1), f2 = f1 & (l2 == ml2);
  u32 mp1 = e2 ? p1 : 0xFFFFFFFFu, mp2 = f2 ? p2 : 0xFFFFFFFFu;
  wave_umin32x2(mp1, mp2);
  c1 = ((u64)mh1 << 32) | ml1; p1 = mp1;
  c2 = ((u64)mh2 << 32) | ml2; p2 = mp2;
}
// N independent argmins in one pass (k_wide's windows: the candidates of N jobs): every DPP step of the N cascades back to back
template <int N>
__device__ __forceinline__ void wave_umin32xN(u32 (&v)[N]) {
#define CNS_DPP_STEPN(ctrl, rmask) _Pragma("unroll") for (int x = 0; x < N; ++x) CNS_DPP_STEP32(op_umin, v[x], 0xFFFFFFFFu, ctrl, rmask)
  CNS_DPP_STEPN(0x111, 0xF);
  CNS_DPP_STEPN(0x112, 0xF);
  CNS_DPP_STEPN(0x114, 0xF);
  CNS_DPP_STEPN(0x118, 0xF);
  CNS_DPP_STEPN(0x142, 0xA);
  CNS_DPP_STEPN(0x143, 0xC);
#undef CNS_DPP_STEPN
#pragma unroll
  for (int x = 0; x < N; ++x) v[x] = rl32(v[x], 63);
}
template <int N>
__device__ __forceinline__ void wave_argminN(u64 (&c)[N], u32 (&p)[N]) {
  u32 mh[N], ml[N], mp[N];
#pragma unroll
  for (int x = 0; x < N; ++x) mh[x] = (u32)(c[x] >> 32);
  wave_umin32xN<N>(mh);
#pragma unroll
  for (int x = 0; x < N; ++x) ml[x] = (u32)(c[x] >> 32) == mh[x] ? (u32)c[x] : 0xFFFFFFFFu;
  wave_umin32xN<N>(ml);
#pragma unroll
  for (int x = 0; x < N; ++x) mp[x] = (((u32)(c[x] >> 32) == mh[x]) & ((u32)c[x] == ml[x])) ? p[x] : 0xFFFFFFFFu;
  wave_umin32xN<N>(mp);
#pragma unroll
  for (int x = 0; x < N; ++x) { c[x] = ((u64)mh[x] << 32) | ml[x]; p[x] = mp[x]; }
}
// the same over 16 per-wave slots replicated in every row
__device__ __forceinline__ void reduce16(u64& c, u32& p) {
  const u32 hi = (u32)(c >> 32), lo = (u32)c;
  const u32 mh = row_umin32(hi);
  const bool e1 = hi == mh;
  const u32 ml = row_umin32(e1 ? lo : 0xFFFFFFFFu);
  const bool e2 = e1 & (lo == ml);
  const u32 mp = row_umin32(e2 ? p : 0xFFFFFFFFu);
  c = ((u64)mh << 32) | ml;
  p = mp;
}

__device__ __forceinline__ int clamp_cpu(i64 c) {
  return c > 0x7FFFFFFFll ? 0x7FFFFFFF : (c < 0 ? 0 : (int)c);
}
__device__ __forceinline__ u32 mem_mib_ceil(u64 m) {
  u64 v = (m + 0xFFFFFull) >> 20;
  if (m > 0xFFFFFFFFFFF00000ull) v = 0xFFFFFFFFull;
  return v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)v;
}
__device__ __forceinline__ u32 nibbles_of(u64 cnt);
__device__ __forceinline__ u32 mem_gib16(u32 mib);
// KParams::dip_*: the packed upper bounds of one time-map entry, and "is it below the front somewhere" (on the summaries:
// what the scanners' filters can tell apart)
__device__ __forceinline__ u32 dip_cm_of(const Res& r) {
  const i64 c = r.cpu <= 0 ? 0 : ((r.cpu + 255) >> 8);
  return ((u32)(c > 0xFFFF ? 0xFFFF : c) << 16) | mem_gib16(mem_mib_ceil(r.mem));
}
__device__ __forceinline__ u64 class_counts(u64 gres, const GresDev& L);
__device__ __forceinline__ bool dip_below(const Res& e, const Res& front, const GresDev& L) {
  if (e.cpu < front.cpu || e.mem < front.mem) return true;
  const u64 ce = class_counts(e.gres, L), cf = class_counts(front.gres, L);
  for (u32 g = 0; g < 8; ++g)
    if (((ce >> (8 * g)) & 0xFFull) < ((cf >> (8 * g)) & 0xFFull)) return true;
  return false;
}
__device__ __forceinline__ u64 class_counts(u64 gres, const GresDev& L) {
  u64 c = 0;
  for (int g = 0; g < (int)L.num_classes; ++g) c |= (u64)popc64(gres & L.class_mask[g]) << (8 * g);
  return c;
}
__device__ __forceinline__ void set_fault(const KParams& P, u32 code, u32 a, u32 b, u32 c) {
  if (atomicCAS(P.fault, 0u, code) == 0u) { P.fault[1] = a; P.fault[2] = b; P.fault[3] = c; }
}
__device__ __forceinline__ Res res_zero() { Res r; r.cpu = 0; r.mem = 0; cores_clear(r); r.gres = 0; return r; }

// The NodeBlock of slot q.  When partitions share nodes (P.slot_block != null) every slot of a node points at the block
// of the node's first slot: ONE time map per craned, whatever partition a job came through (JobScheduler.cpp:6563,6609-6617).
__device__ __forceinline__ NodeHdr* hdr_raw(const KParams& P, u32 q) { return (NodeHdr*)(P.blocks + (u64)q * P.block_stride); }
__device__ __forceinline__ NodeHdr* hdr_of(const KParams& P, u32 q) {
  if (P.slot_block) q = P.slot_block[q];
  return hdr_raw(P, q);
}
template <bool kW = true>   // kW = false: the caller knows that the snapshot has no core id above 127 (see load_block)
__device__ __forceinline__ TlMap tl_of(const KParams& P, const NodeHdr* h) {
  TlMem* m = (TlMem*)((char*)h + sizeof(NodeHdr));
  return TlMap{m, (TlExt*)(m + P.tl_cap), kW ? P.wide_cores : 0u};
}

// ---------------------------------------------------------------------------------------------
// k_init_nodes — prologue
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_init_nodes(const KParams* __restrict__ Pp) {
  // NodeState::InitTimeAvailResMap (JobScheduler.h:301-338) + the NodeRater constructor (h:498-511) of one slot:
  // a partition's node, or the virtual node of a reservation (res_total = the reserved share, map ends at the
  // reservation's end).  Reservation entries of a real node (cpp:6627-6679): expired ones are ignored, an ACTIVE
  // one is an allocation until its end, a FUTURE one is a dip [start, end) (h:305-308).
  const KParams& P = *Pp;
  u32 q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= P.num_slots) return;
  const u32 n = P.slot_node[q];
  const Res tot = P.slot_total[q];
  const i64 end_h = P.slot_end[q];
  Res a0 = tot;
  double cost = 0.0;
  NodeHdr* hd = hdr_raw(P, q);      // (a secondary slot of a shared node builds a private copy nobody reads: same values)
  const TlMap T = tl_of(P, hd);       // T[i].t / T[i].r: change times and what is RELEASED there
  const TlMap A2 = T + kTlCap / 2;     // A2[i].r: what is ALLOCATED at T[i].t (only with reservations; host bounds the count)
  u32 len = 1;  // T[0] reserved for {now, avail0}
  const double tcpu = (double)tot.cpu / 256.0;
  const u32 rvb = P.rv_off[q], rve = P.rv_off[q + 1];
  const bool has_rv = rve > rvb;
  // sorted insert of a change at `time`: kind 0 = release (+r), 1 = allocation (-r); equal times accumulate,
  // at one time releases apply before allocations (h:315-320)
  auto put = [&](i64 time, const Res& r, int kind) {
    u32 i = 1;
    while (i < len && T[i].t() < time) ++i;
    if (!(i < len && T[i].t() == time)) {
      for (u32 m = len; m > i; --m) { T[m] = T[m - 1]; if (has_rv) A2[m] = A2[m - 1]; }
      TlEntry z; z.t = time; z.r = res_zero();
      T[i] = z;
      if (has_rv) A2[i] = z;
      ++len;
    }
    const TlSlot dst = kind == 0 ? T[i] : A2[i];
    Res acc = dst.r();
    res_add(acc, r);
    dst.set_r(acc);
  };
  i64 first = kInf;
  for (u32 a = rvb; a < rve; ++a) {  // NodeRater: reserved_res first (h:502-506)
    const i64 st = P.rv_start[a], en = P.rv_end[a];
    if (P.now >= en) continue;       // expired (cpp:6631-6634)
    first = st < first ? st : first; // craned_id_first_resv_map (cpp:6635-6642)
    if (P.now < st) {
      const double ratio = ((double)P.rv_res[a].cpu / 256.0) / tcpu;
      cost += (double)(en - st) * ratio;
    }
  }
  for (u32 a = rvb; a < rve; ++a) {
    const i64 st = P.rv_start[a], en = P.rv_end[a];
    if (P.now >= en) continue;
    const Res r = P.rv_res[a];
    if (P.now >= st) {               // active: allocated until its end (cpp:6644-6652), before the running jobs
      res_sub(a0, r);
      const double ratio = ((double)r.cpu / 256.0) / tcpu;
      cost += (double)(en - P.now) * ratio;
      put(en, r, 0);
    } else {                         // future: dip (cpp:6669-6677)
      put(st, r, 1);
      put(en, r, 0);
    }
  }
  for (u32 a = P.rn_off[q]; a < P.rn_off[q + 1]; ++a) {
    i64 end = P.rn_end[a];
    if (end < P.now + 1) end = P.now + 1;  // JobScheduler.cpp:6513-6514
    const Res r = P.rn_res[a];
    res_sub(a0, r);                        // JobScheduler.h:313
    // NodeRater ctor, JobScheduler.h:508-510 + MinCpuTimeRatioFirst :47-53 (ratio first, then x secs)
    double ratio = ((double)r.cpu / 256.0) / tcpu;
    cost += (double)(end - P.now) * ratio;
    put(end, r, 0);                        // release {end, r}; equal end times accumulate (JobScheduler.h:325-335)
  }
  { TlEntry e0; e0.t = P.now; e0.r = a0; T[0] = e0; }
  for (u32 i = 1; i < len; ++i) {  // value at a change time = previous value + releases - allocations
    Res v = T[i - 1].r();
    res_add(v, T[i].r());
    if (has_rv) res_sub(v, A2[i].r());
    T[i].set_r(v);
  }
  {  // time_avail_res_map[end].SetToZero(), JobScheduler.h:337 (end = InfiniteFuture, or the reservation's end)
    u32 i = 1;
    while (i < len && T[i].t() < end_h) ++i;
    if (!(i < len && T[i].t() == end_h)) {
      for (u32 m = len; m > i; --m) T[m] = T[m - 1];
      ++len;
    }
    { TlEntry z; z.t = end_h; z.r = res_zero(); T[i] = z; }
  }
  hd->len = len;
  if (P.f_len) P.f_len[q] = len;
  hd->node = n;
  hd->type = P.slot_type[q];
  hd->pad = 0;
  hd->avail0 = a0;
  hd->total = tot;
  P.cost[q] = cost;
  P.f_cpu[q] = clamp_cpu(a0.cpu);
  P.f_mem[q] = mem_mib_ceil(a0.mem);
  P.f_cnt[q] = class_counts(a0.gres, P.gres);
  {  // the earliest entry below the front (a pending reservation's dip; the zero entry at the end of a reservation's own map)
    u32 dt = 0xFFFFFFFFu, dcm = 0, dg = 0;
    for (u32 i = 1; i < len; ++i) {
      const TlEntry ei = T[i];
      if (ei.t == kInf || ei.t - P.now >= 0xFFFFFFFFll) break;
      if (dip_below(ei.r, a0, P.gres)) { dt = (u32)(ei.t - P.now); dcm = dip_cm_of(ei.r); dg = nibbles_of(class_counts(ei.r.gres, P.gres)); break; }
    }
    P.dip_t[q] = dt; P.dip_cm[q] = dcm; P.dip_g[q] = dg;
  }
  P.first_resv[q] = first;
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
  u32 orig;      // index in the caller's queue
  u64 poff;      // first placement record of the job
};


// Job records are 32-dword AoS rows in HBM; lane i of a wave loads dword i, so a whole record costs ONE
// vector register while it is in flight (it is fetched one job ahead) and its fields are then read out
// with v_readlane straight into scalar registers.
enum JobRecField : u32 {
  kJrL = 0, kJrNcpu = 2, kJrNmem = 4, kJrTcpu = 6, kJrTmem = 8, kJrGspec = 10, kJrK = 12, kJrNtasks = 13,
  kJrTmin = 14, kJrTmax = 15, kJrFlags = 16, kJrGtot = 17, kJrOrig = 18, kJrPoff = 20, kJrInclB = 22,
  kJrInclE = 24, kJrExclB = 26, kJrExclE = 28,
  // derived by k_prep_jobs
  kJdMcpu = 32,   // minimum view = req_node + req_task * tpn_min (JobScheduler.cpp:6154-6156): cpu (i64)
  kJdMmem = 34,   //                                                                               mem (u64)
  kJdRc32 = 36,   // min-view cpu clamped to the scanners' 32-bit front summary
  kJdRm16 = 37,   // min-view mem in GiB, rounded down, saturating at 0xFFFF
  kJdRq = 38,     // specified GRES counts per class as nibbles, saturating at 15 (like the node side)
  kJdShape = 39,  // bit 0: ntasks != node_num, bit 1: tpn_min == 1, bit 2: satisfiable under the 32-bit summaries,
                  // bit 3: untyped GRES of a name with several classes (which classes serve it depends on the node, :577-592)
  kJdGmode = 40,  // GRES request shape: 0 none; bit 0 one specified class, bit 1 one untyped total; 4 general
  kJdGsel = 41,   // nibble shift of that class | index of that name << 8
  kJdGneed = 42,  // its count | the total << 8, both saturated at 15
  kJdTyok = 44,   // u64: bit t = the minimum view fits res_total of node type t (:6171-6175, :6222-6223)
  kJdAcnt = 46,   // GRES slots per class an allocation of the minimum view takes (nibbles, saturating at 15) — only
                  // meaningful when kJdShape bit 3 is clear (the split over the classes does not depend on the node)
  // k_pipe's scanners read everything they need from 12 consecutive dwords (12 v_readlane instead of 35):
  kJsHead = 48,   // flags | min(node_num, 0xFFFF) << 8 | shape << 24
  kJsRc32 = 49, kJsMem = 50 /* rm16 | dm16 << 16 (min-view mem in whole GiB, rounded down) */, kJsRq = 51, kJsGtot = 52,
  kJsGres = 53,   // gmode | gsel << 4 | gneed << 16
  kJsL = 54, kJsAcnt = 56, kJsTyok = 58
};
// ---------------------------------------------------------------------------------------------
// k_pack_jobs: the caller's job SoA (uploaded as it is) -> dwords 0..29 of the 64-dword job records, grouped by
// partition in queue order (`grouped[i]` = caller index of the i-th grouped job; the grouping itself is a counting
// pass over one u32 per job on the host).  One thread per job; replaces a 256-byte-per-job staging loop on the host.
// ---------------------------------------------------------------------------------------------
struct PackParams {
  u64 Jg;
  const u32* grouped;
  const i64* L; const i64* ncpu; const u64* nmem; const i64* tcpu; const u64* tmem;
  const u32* k; const u32* ntasks; const u32* tmin; const u32* tmax;
  const uint8_t* excl; const uint8_t* gtot; const uint8_t* gspec;
  const u64* incl_off; const u64* excl_off; const u64* place_off;
  u32* jobrec;
  const uint8_t* tag;   // member tag of the job's partition inside its group (null: no partitions share nodes)
};
__global__ __launch_bounds__(256) void k_pack_jobs(const PackParams P) {
  const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  if (i >= P.Jg) return;
  const u64 j = P.grouped[i];
  u32* rec = P.jobrec + i * kJobRecDwords;
  auto put64 = [&](u32 f, u64 v) { rec[f] = (u32)v; rec[f + 1] = (u32)(v >> 32); };
  u32 flags = 0, gt = 0;
  u64 gs = 0;
  if (P.gtot) { const uint8_t* g = P.gtot + j * 4; gt = g[0] | (u32)g[1] << 8 | (u32)g[2] << 16 | (u32)g[3] << 24; }
  if (P.gspec) { const uint8_t* g = P.gspec + j * 8; for (u32 c = 0; c < 8; ++c) gs |= (u64)g[c] << (8 * c); }
  if (gt | gs) flags |= kJfGres;
  if (P.excl && P.excl[j]) flags |= kJfExclusive;
  if (P.tag) flags |= (u32)P.tag[j] << 8;
  u64 ib = 0, ie = 0, eb = 0, ee = 0;
  if (P.incl_off) { ib = P.incl_off[j]; ie = P.incl_off[j + 1]; if (ie > ib) flags |= kJfIncl; else ie = ib; }
  if (P.excl_off) { eb = P.excl_off[j]; ee = P.excl_off[j + 1]; if (ee > eb) flags |= kJfExcl; else ee = eb; }
  put64(kJrL, (u64)P.L[j]);
  put64(kJrNcpu, (u64)(P.ncpu ? P.ncpu[j] : 0));
  put64(kJrNmem, P.nmem[j]);
  put64(kJrTcpu, (u64)P.tcpu[j]);
  put64(kJrTmem, P.tmem[j]);
  put64(kJrGspec, gs);
  rec[kJrK] = P.k[j]; rec[kJrNtasks] = P.ntasks[j]; rec[kJrTmin] = P.tmin[j]; rec[kJrTmax] = P.tmax[j];
  rec[kJrFlags] = flags; rec[kJrGtot] = gt; rec[kJrOrig] = (u32)j; rec[19] = 0;
  put64(kJrPoff, P.place_off[j]);
  put64(kJrInclB, ib); put64(kJrInclE, ie); put64(kJrExclB, eb); put64(kJrExclE, ee);
  rec[30] = 0; rec[31] = 0;
}

__device__ __forceinline__ u32 fetch_job(const KParams& P, u64 ji) {
  return *as_global(P.jobrec + (ji * kJobRecDwords + (threadIdx.x & (kJobRecDwords - 1))));
}
__device__ __forceinline__ u64 jr64(u32 raw, u32 f) { return ((u64)rl32(raw, f + 1) << 32) | rl32(raw, f); }
__device__ __forceinline__ JobCtx make_job(const KParams& P, u64 ji, u32 raw) {
  JobCtx J;
  J.ji = ji;
  J.L = (i64)jr64(raw, kJrL);
  J.E = P.now + J.L;
  J.node_view.cpu = (i64)jr64(raw, kJrNcpu);
  J.node_view.mem = jr64(raw, kJrNmem);
  J.node_view.gtot = rl32(raw, kJrGtot);
  J.node_view.gspec = jr64(raw, kJrGspec);
  J.tcpu = (i64)jr64(raw, kJrTcpu);
  J.tmem = jr64(raw, kJrTmem);
  J.k = rl32(raw, kJrK);
  J.ntasks = rl32(raw, kJrNtasks);
  J.tmin = rl32(raw, kJrTmin);
  J.tmax = rl32(raw, kJrTmax);
  J.flags = rl32(raw, kJrFlags);
  J.general = J.ntasks != J.k;
  J.min_view = compose(J.node_view, J.tcpu, J.tmem, J.tmin);
  J.incl_b = jr64(raw, kJrInclB); J.incl_e = jr64(raw, kJrInclE);
  J.excl_b = jr64(raw, kJrExclB); J.excl_e = jr64(raw, kJrExclE);
  J.orig = rl32(raw, kJrOrig);
  J.poff = jr64(raw, kJrPoff);
  return J;
}

// Compact view of a job for the worker's inline fast path (the full JobCtx only exists in LDS, for the
// out-of-line multi-node / general paths): keeps the worker's scalar-register footprint small.
struct FastJob {
  i64 L, E;
  Req mv;        // minimum view = req_node + req_task * tpn_min
  u32 flags, k, tmin, ntasks;
  u32 orig;
  u64 poff;
};
__device__ __forceinline__ FastJob make_fast_job(const KParams& P, u32 raw) {
  FastJob F;
  F.L = (i64)jr64(raw, kJrL);
  F.E = P.now + F.L;
  F.k = rl32(raw, kJrK);
  F.ntasks = rl32(raw, kJrNtasks);
  F.tmin = rl32(raw, kJrTmin);
  F.flags = rl32(raw, kJrFlags);
  F.mv.cpu = (i64)jr64(raw, kJdMcpu); F.mv.mem = jr64(raw, kJdMmem);  // composed by k_prep_jobs
  F.mv.gtot = rl32(raw, kJrGtot); F.mv.gspec = jr64(raw, kJrGspec);
  F.orig = rl32(raw, kJrOrig);
  F.poff = jr64(raw, kJrPoff);
  return F;
}
// full decode into LDS for the out-of-line paths
__device__ __noinline__ void job_to_lds(const KParams& P, u64 ji, u32 raw, JobCtx* dst) {
  const JobCtx J = make_job(P, ji, raw);
  if ((threadIdx.x & 63u) == 0) *dst = J;
  __threadfence_block();
}
__device__ __forceinline__ bool req_impossible(const Req& mv) {
  return mv.cpu > 0x7FFFFFFEll || (mv.gspec & 0x8080808080808080ull) != 0;
}

// membership of node n in the job's included / excluded list (JobScheduler.cpp:6202-6220)
__device__ __forceinline__ bool in_list(const u32* lst, u64 b, u64 e, u32 n) {
  for (u64 i = b; i < e; ++i)
    if (lst[i] == n) return true;
  return false;
}


// Per-type static data held by lane t: exact cpu / mem, number of core ids, per-class slot counts.
struct TypeLane { i64 cpu; u64 mem; u32 ncores; u64 cnt; };

// ---------------------------------------------------------------------------------------------
// k_prep_jobs: job-parallel pre-pass (one thread per pending job of the cycle).  Everything about a job
// that does not depend on the evolving node state is computed here, off the sequential chain: the
// minimum view, the request side of the scanners' filters, the shape of the GRES request and the
// "fits res_total" mask over the node types.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 nibbles_of(u64 cnt);
__global__ __launch_bounds__(256) void k_prep_jobs(const KParams* __restrict__ Pp, u64 njobs) {
  const KParams& P = *Pp;
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= njobs) return;
  u32* rec = const_cast<u32*>(P.jobrec) + i * kJobRecDwords;
  auto g64 = [&](u32 f) { return ((u64)rec[f + 1] << 32) | rec[f]; };
  Req nv;
  nv.cpu = (i64)g64(kJrNcpu); nv.mem = g64(kJrNmem); nv.gtot = rec[kJrGtot]; nv.gspec = g64(kJrGspec);
  const u32 k = rec[kJrK], ntasks = rec[kJrNtasks], tmin = rec[kJrTmin];
  u32 flags = rec[kJrFlags] & ~(u32)kJfMayPreempt;
  if (P.pre.enabled) {   // TryPreempt_ returns at once for a job whose qos may preempt nobody (:6384-6385): such a job of a cycle with
    const u32 jq = P.pre.pj_qos[rec[kJrOrig]];   // preemption keeps k_select's inline path (the worker's and the scanners' `shared_nodes`)
    if (jq < P.pre.num_qos && P.pre.qp_off[jq] != P.pre.qp_off[jq + 1]) flags |= kJfMayPreempt;
  }
  rec[kJrFlags] = flags;
  const Req mv = compose(nv, (i64)g64(kJrTcpu), g64(kJrTmem), tmin);
  const bool possible = !req_impossible(mv);
  const u32 rq = nibbles_of(nv.gspec);
  u32 gmode = 0, gsel = 0, gneed = 0;
  if (flags & kJfGres) {
    // most requests name ONE class ("gpu:a100:2") and / or ONE untyped total ("gpu:2"): those get a
    // 3- / 8-instruction test per node in the scan instead of the general one
    u32 nspec = 0, nname = 0, g0 = 0, a0 = 0;
    for (u32 g = 0; g < 8; ++g)
      if ((nv.gspec >> (8 * g)) & 0xFFull) { ++nspec; g0 = g; }
    for (u32 a = 0; a < (u32)kMaxNames; ++a)
      if ((nv.gtot >> (8 * a)) & 0xFFu) { ++nname; a0 = a; }
    gmode = 4;
    if (nspec <= 1 && nname <= 1) {
      const u32 tot = (nv.gtot >> (8 * a0)) & 0xFFu, spec = (u32)((nv.gspec >> (8 * g0)) & 0xFFull);
      gmode = (nspec ? 1u : 0u) | (nname ? 2u : 0u);
      // a total that the specified class of the same name already covers adds nothing to the filter
      if (gmode == 3 && ((P.gres.name_bytes[a0] >> (8 * g0)) & 0xFFull) != 0 && tot <= spec) gmode = 1;
      gsel = (4 * g0) | (a0 << 8);
      gneed = ((rq >> (4 * g0)) & 15u) | ((tot > 15u ? 15u : tot) << 8);
    }
  }
  // per-class slot counts of GetFeasibleResourceInNode's allocation (:556-592) when they do not depend on the node
  u32 acnt = 0;
  bool dyn_gres = false;
  if (flags & kJfGres) {
    u64 cnt = nv.gspec;
    for (u32 a = 0; a < (u32)kMaxNames; ++a) {
      const u32 tot = (nv.gtot >> (8 * a)) & 0xFFu;
      const u64 nb = P.gres.name_bytes[a];
      const u32 ssum = byte_sum(nv.gspec & nb);
      if (tot <= ssum) continue;
      u32 ncls = 0, g0 = 0;
      for (u32 g = 0; g < 8; ++g)
        if ((nb >> (8 * g)) & 0xFFull) { ++ncls; g0 = g; }
      if (ncls == 1) cnt += (u64)(tot - ssum) << (8 * g0);
      else dyn_gres = true;
    }
    acnt = nibbles_of(cnt);
  }
  u64 tyok = 0;
  for (u32 t = 0; t < P.num_types; ++t) {
    const Res tt = P.type_total[t];
    if (feasible_counts(mv, tt.cpu, tt.mem, cores_count(tt), class_counts(tt.gres, P.gres), P.gres))
      tyok |= 1ull << t;
  }
  rec[kJdMcpu] = (u32)(u64)mv.cpu; rec[kJdMcpu + 1] = (u32)((u64)mv.cpu >> 32);
  rec[kJdMmem] = (u32)mv.mem; rec[kJdMmem + 1] = (u32)(mv.mem >> 32);
  rec[kJdRc32] = (u32)(mv.cpu > 0x7FFFFFFFll ? 0x7FFFFFFF : (int)mv.cpu);
  rec[kJdRm16] = (mv.mem >> 30) > 0xFFFFull ? 0xFFFFu : (u32)(mv.mem >> 30);
  rec[kJdRq] = rq;
  rec[kJdShape] = (ntasks != k ? 1u : 0u) | (tmin == 1 ? 2u : 0u) | (possible ? 4u : 0u) | (dyn_gres ? 8u : 0u);
  rec[kJdAcnt] = acnt;
  {
    const u64 mgib = mv.mem >> 30;
    const u32 rm16 = mgib > 0xFFFFull ? 0xFFFFu : (u32)mgib;
    const u32 shape = (ntasks != k ? 1u : 0u) | (tmin == 1 ? 2u : 0u) | (possible ? 4u : 0u) | (dyn_gres ? 8u : 0u);
    rec[kJsHead] = (flags & 0xFFu) | ((k > 0xFFFFu ? 0xFFFFu : k) << 8) | (shape << 24);
    rec[kJsRc32] = (u32)(mv.cpu > 0x7FFFFFFFll ? 0x7FFFFFFF : (int)mv.cpu);
    rec[kJsMem] = rm16 | (rm16 << 16);
    rec[kJsRq] = rq; rec[kJsGtot] = nv.gtot;
    rec[kJsGres] = gmode | (gsel << 4) | (gneed << 16);
    rec[kJsL] = rec[kJrL]; rec[kJsL + 1] = rec[kJrL + 1];
    rec[kJsAcnt] = acnt; rec[kJsAcnt + 1] = 0;
    rec[kJsTyok] = (u32)tyok; rec[kJsTyok + 1] = (u32)(tyok >> 32);
  }
  rec[kJdGmode] = gmode; rec[kJdGsel] = gsel; rec[kJdGneed] = gneed;
  rec[kJdTyok] = (u32)tyok; rec[kJdTyok + 1] = (u32)(tyok >> 32);
}

// slot code = row << 10 | t, t = scanner lane 0..kScan-1; partition-local slot = row * kScan + t
template <u32 kS> __device__ __forceinline__ u32 slot_of_code_t(u32 code) { return (code >> 10) * kS + (code & 1023u); }
__device__ __forceinline__ u32 slot_of_code(u32 code) { return slot_of_code_t<kScan>(code); }
// wave that owns the slot: scanner index t / 64 -> waves 1,2,3,5,6,7 (wave kIdleWave holds no nodes)
__device__ __forceinline__ u32 owner_wave(u32 code) {
  const u32 sid = (code & 1023u) >> 6;
  return sid + 1u + (sid + 1u >= (u32)kIdleWave ? 1u : 0u);
}

// ---------------------------------------------------------------------------------------------
// worker routines (all 64 lanes of wave 0 execute them)
// ---------------------------------------------------------------------------------------------

// Window-min over [now, E) from a register-resident chunk (lane i holds entry i, len <= 64):
// min_res_on_node = res_avail; for entries with time < E: Ckmin  (JobScheduler.cpp:6278-6283).
__device__ __forceinline__ Res window_min_regs(const TlEntry& e, bool act, const Res& a0, i64 E) {
  const bool inw = act && e.t < E;
  i64 cpu = inw ? e.r.cpu : a0.cpu;
  u64 mem = inw ? e.r.mem : a0.mem;
  const bool hasc = inw && !cores_empty(e.r);  // an empty core set is skipped by Ckmin
  u64 clo = hasc ? e.r.clo : ~0ull, chi = hasc ? e.r.chi : ~0ull, c2 = hasc ? e.r.c2 : ~0ull, c3 = hasc ? e.r.c3 : ~0ull;
  u64 g = inw ? e.r.gres : ~0ull;
  Res m;
  cpu = wave_min_i64(cpu);
  m.cpu = cpu < a0.cpu ? cpu : a0.cpu;
  mem = wave_min_u64(mem);
  m.mem = mem < a0.mem ? mem : a0.mem;
  clo = wave_and_u64(clo);
  chi = wave_and_u64(chi);
  m.gres = wave_and_u64(g) & a0.gres;
  if (a0.c2 | a0.c3) { c2 = wave_and_u64(c2); c3 = wave_and_u64(c3); }   // (uniform: a node without core ids above 127 pays nothing)
  if (!cores_empty(a0)) { m.clo = a0.clo & clo; m.chi = a0.chi & chi; m.c2 = a0.c2 & c2; m.c3 = a0.c3 & c3; }
  else cores_clear(m);
  return m;
}

// General form for time maps longer than one chunk.
__device__ __noinline__ Res window_min(const TlMap T, u32 len, const Res& a0, i64 E, u32 lane) {
  i64 cpu = a0.cpu;
  u64 mem = a0.mem, clo = ~0ull, chi = ~0ull, c2 = ~0ull, c3 = ~0ull, g = a0.gres;
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
      if (!cores_empty(e.r)) { clo &= e.r.clo; chi &= e.r.chi; c2 &= e.r.c2; c3 &= e.r.c3; }
      g &= e.r.gres;
    }
    if (__any(act && !inw)) break;  // sorted by time: nothing later is inside the window
  }
  Res m;
  m.cpu = wave_min_i64(cpu);
  m.mem = wave_min_u64(mem);
  clo = wave_and_u64(clo);
  chi = wave_and_u64(chi);
  c2 = wave_and_u64(c2);
  c3 = wave_and_u64(c3);
  m.gres = wave_and_u64(g);
  if (!cores_empty(a0)) { m.clo = a0.clo & clo; m.chi = a0.chi & chi; m.c2 = a0.c2 & c2; m.c3 = a0.c3 & c3; }
  else cores_clear(m);
  return m;
}

// Exclusive job: every entry with time < E must still hold the whole node (JobScheduler.cpp:6249-6257).
__device__ __noinline__ bool window_all_total(const TlMap T, u32 len, const Res& tot, i64 E, u32 lane) {
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
// end boundary copies the un-subtracted value of the entry covering `end`).
// Register form: lane i holds entry i of a map with len <= 64.  Returns the new length.
__device__ __forceinline__ u32 tl_commit_regs(const KParams& P, NodeHdr* hd, const TlMap T, TlEntry e, u32 len,
                                              i64 start, i64 end, const Res& res, u32 lane, u32 job) {
  const bool act = lane < len;
  const i64 t = act ? e.t : kInf;
  const u32 c_start = (u32)__popcll(__ballot(act && t <= start));
  const u32 c_end = (u32)__popcll(__ballot(act && t <= end));
  if (c_start == 0 || c_end > len || (c_end == len && rl64((u64)e.t, len - 1) != (u64)end)) {
    // cases #1/#2 cannot occur: the last entry (INF, or the end of a reservation's map) is a zero entry, so no
    // feasible window reaches past it; a job may END exactly there
    if (lane == 0) set_fault(P, 1, job, hd->node, len);
    return len;
  }
  const u32 ib = c_start - 1, ie0 = c_end - 1;
  TlEntry eb, ee;
  eb.t = (i64)rl64((u64)e.t, ib);
  ee.t = (i64)rl64((u64)e.t, ie0);
  eb.r = res_zero();
  ee.r = res_zero();
  const bool ins_s = eb.t != start, ins_e = ee.t != end;
  const u32 np = lane + ((ins_s && lane > ib) ? 1u : 0u) + ((ins_e && lane > ie0) ? 1u : 0u);
  const bool sub = act && t >= start && t < end;
  if (ins_s) eb.r = rl_res(e.r, ib);    // pre-subtraction values of the covering entries
  if (ins_e) ee.r = rl_res(e.r, ie0);
  if (sub) res_sub(e.r, res);
  if (act && lane >= ib && (np != lane || sub)) T[np] = e;
  const u32 nlen = len + (ins_s ? 1u : 0u) + (ins_e ? 1u : 0u);
  if (lane == 0) {
    if (ins_s) { TlEntry s = eb; s.t = start; res_sub(s.r, res); T[ib + 1] = s; }
    if (ins_e) { TlEntry x = ee; x.t = end; T[ie0 + (ins_s ? 1u : 0u) + 1] = x; }
    hd->len = nlen;
  }
  return nlen;
}

// General form (any length): chunks are rewritten from the highest down so that the <= 2-slot shift
// never overwrites an entry that has not been read yet.
template <bool kRelease = false>   // kRelease: UpdateResourceInNode(..., is_release = true): += instead of -=
__device__ __noinline__ u32 tl_commit(const KParams& P, NodeHdr* hd, i64 start, i64 end, const Res& res, u32 lane, u32 job) {
  const TlMap T = tl_of(P, hd);
  const u32 len = hd->len;
  u32 c_start = 0, c_end = 0;  // #entries with t <= start / t <= end
  for (u32 base = 0; base < len; base += 64) {
    u32 i = base + lane;
    bool act = i < len;
    i64 t = act ? T[i].t() : kInf;
    c_start += __popcll(__ballot(act && t <= start));
    c_end += __popcll(__ballot(act && t <= end));
    if (__any(act && t > end)) break;
  }
  if (c_start == 0 || c_end > len || (c_end == len && T[len - 1].t() != end) || len + 2 > P.tl_cap) {
    if (lane == 0) set_fault(P, 1, job, hd->node, len);
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
    if (sub) { if (kRelease) res_add(e.r, res); else res_sub(e.r, res); }
    if (act && (np != i || sub)) T[np] = e;
  }
  const u32 nlen = len + (ins_s ? 1u : 0u) + (ins_e ? 1u : 0u);
  if (lane == 0) {
    if (ins_s) { TlEntry s = eb; s.t = start; if (kRelease) res_add(s.r, res); else res_sub(s.r, res); T[ib + 1] = s; }
    if (ins_e) { TlEntry x = ee; x.t = end; T[ie0 + (ins_s ? 1u : 0u) + 1] = x; }
    hd->len = nlen;
  }
  __threadfence_block();
  return nlen;
}


// Wave-parallel form for time maps of any length: 64 entries per step, "alloc fits" ballot per chunk, run
// state (inside a satisfied run? its start) carried across chunks.  All lanes return the same value.
__device__ __noinline__ i64 next_fit_wave(const TlMap T, u32 len, const Res* alloc_p, i64 L, i64 t0) {
  const u32 lane = threadIdx.x & 63u;
  const Res alloc = *alloc_p;
  bool in_run = false;
  i64 s = t0;
  for (u32 base = 0; base < len; base += 64) {
    const u32 i = base + lane;
    const bool act = i < len;
    TlEntry e;
    e.t = kInf;
    e.r = res_zero();
    if (act) e = T[i];
    const u32 n = len - base < 64 ? len - base : 64;
    const u64 valid = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
    const u64 sat = __ballot(act && res_le(alloc, e.r));
    // The first relevant entry is the one covering t0 = the last entry with t <= t0 (T[0].t = now <= t0).
    const u32 c = (u32)__popcll(__ballot(act && e.t <= t0));
    u32 pos = 0;  // first bit of this chunk still to be looked at
    if (c != 0) {
      if (c == n && base + 64 < len && T[base + 64].t() <= t0) continue;  // the covering entry is further on
      pos = c - 1;
    }
    while (pos < n) {
      const u64 from = ~0ull << pos;
      if (!in_run) {
        const u64 cand = sat & valid & from;
        if (cand == 0) break;
        const u32 ns = (u32)__builtin_ctzll(cand);
        const i64 tn = (i64)rl64((u64)e.t, ns);
        s = tn > t0 ? tn : t0;
        in_run = true;
        pos = ns + 1;
      } else {
        const u64 un = ~sat & valid & from;
        if (un == 0) break;
        const u32 ue = (u32)__builtin_ctzll(un);
        const i64 endt = (i64)rl64((u64)e.t, ue);
        if (endt - s >= L) return s;   // kth_time + time_limit <= next flip (JobScheduler.h:837-839)
        in_run = false;
        pos = ue + 1;
      }
    }
    if (in_run && (i64)rl64((u64)e.t, n - 1) - s >= L) return s;  // the run already covers [s, s + L)
  }
  return in_run ? s : kInf;  // satisfied through the last entry ("ReachEnd"), or never
}

// Same question for ONE node whose map sits in registers (lane i = entry i, len <= 64), answered with
// a ballot of "alloc fits entry i" and bit scans over its runs.
__device__ __forceinline__ i64 next_fit_regs(const TlEntry& e, u32 len, const Res& alloc, i64 L, i64 t0, u32 lane) {
  // EarliestStartSubsetSelector on one node (JobScheduler.h:792-865) without its walk: every lane decides for ITS
  // entry whether a run of satisfied entries starts there (the entry covering t0 counts with start t0) and whether
  // that run lasts L — the end of the run is the first unsatisfied entry above it, fetched across lanes in one
  // ds_bpermute — and the lowest such lane wins.  (The walk over the runs cost ~150 cycles per run on the worker's
  // serial chain: 4.9 k cycles per backfilled job on C4's deep time maps.)
  const bool act = lane < len;
  const u64 valid = len >= 64 ? ~0ull : ((1ull << len) - 1ull);
  const u64 sat = __ballot(act && res_le(alloc, e.r));
  const u32 idx0 = (u32)__popcll(__ballot(act && e.t <= t0)) - 1u;  // entry covering t0 (T[0].t = now <= t0)
  const u64 unsat = ~sat & valid;
  const bool here = ((sat >> lane) & 1ull) != 0;
  const bool below = lane > 0 && ((sat >> (lane - 1u)) & 1ull) != 0;
  const bool starts = here && (lane == idx0 || (lane > idx0 && !below));
  const i64 s = lane == idx0 ? t0 : e.t;
  const u64 above = lane >= 63 ? 0ull : (unsat & ~((2ull << lane) - 1ull));
  const u32 ue = above ? (u32)__builtin_ctzll(above) : lane;
  const u32 elo = (u32)__shfl((int)(u32)(u64)e.t, (int)ue), ehi = (u32)__shfl((int)(u32)((u64)e.t >> 32), (int)ue);
  const i64 endt = (i64)(((u64)ehi << 32) | elo);
  const u64 ok = __ballot(starts && (above == 0 || endt - s >= L));
  if (!ok) return kInf;
  const u32 w = (u32)__builtin_ctzll(ok);
  return w == idx0 ? t0 : (i64)rl64((u64)e.t, w);
}

// Shared by the "start now" and "backfill" endings of the general path: H[0..k) holds the selected
// nodes with their assigned task counts and allocations; commits them into the time maps and costs,
// emits the placement records (sorted by node index) and the owner updates.
template <u32 kS = kScan>
__device__ __noinline__ void commit_selection(const KParams& P, const JobCtx& J, HeapEnt* H, u32 qbeg, i64 start,
                                 u32 lane, UpdRec* s_upd, int* s_nupd) {
  const i64 end = start + J.L;  // job->end_time = start_time + time_limit, JobScheduler.cpp:6772
  const u32 orig = J.orig;
  const u64 poff = J.poff;
  u32 nup = J.k;
  for (u32 i = 0; i < J.k; ++i) {
    HeapEnt ent = H[i];
    const u32 q = qbeg + slot_of_code_t<kS>(ent.p);
    NodeHdr* hd = hdr_of(P, q);
    const Res tot = hd->total;
    const Res e0 = tl_of(P, hd)[0].r();
    u32 newlen = tl_commit(P, hd, start, end, ent.res, lane, orig);
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
      P.cost[q] = ncost;
      if (u.has_front) { P.f_cpu[q] = u.fcpu; P.f_mem[q] = u.fmem; P.f_cnt[q] = u.fcnt; }
      if (P.f_len) P.f_len[q] = newlen;
      s_upd[i] = u;
    }
    if (P.sib_off) {
      // the node's slots in the OTHER partitions of the group see the same time map: new length and front summary,
      // their own cost untouched (NodeRater.cost is per selector, JobScheduler.h:498-516); listed after the k own records
      for (u32 a = P.sib_off[q]; a < P.sib_off[q + 1]; ++a) {
        const u32 qs = P.sib[a];
        if (lane == 0) {
          UpdRec us = u;
          const u32 ps = qs - qbeg;
          us.p = ((ps / kS) << 10) | (ps % kS);
          us.has_front = u.has_front | 2u;
          if (u.has_front) { P.f_cpu[qs] = u.fcpu; P.f_mem[qs] = u.fmem; P.f_cnt[qs] = u.fcnt; }
          if (P.f_len) P.f_len[qs] = newlen;
          s_upd[nup] = us;
        }
        ++nup;
      }
    }
  }
  if (lane == 0) *s_nupd = (int)nup;
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
    if (P.o_c2) { P.o_c2[o] = me.res.c2; P.o_c3[o] = me.res.c3; }
    if (P.pre.enabled) {   // UpdateNodeSelectorWithScheduledJob (h:636-642): the job joins its nodes' qos_job_map
      const u32 q = qbeg + slot_of_code_t<kS>(me.p);
      P.pre.rec_orig[o] = orig; P.pre.rec_slot[o] = q; P.pre.rec_gone[o] = 0;
      const u32 hq = P.slot_block ? P.slot_block[q] : q;   // the list is the NODE's (one NodeState per craned)
      P.pre.rec_next[o] = P.pre.slot_head[hq];   // (the k nodes are distinct: no two lanes touch one list)
      P.pre.slot_head[hq] = (u32)o;
    }
  }
  if (P.pre.enabled && lane == 0) { P.pre.pj_rec0[orig] = (u32)poff; P.pre.pj_k[orig] = J.k; P.pre.pj_end[orig] = end; }
}

// Single-node ending (node_num == 1) straight from the register-resident chunk: commit, cost, owner
// update and the one placement record.
// What the scanners (and the worker's own merge) know about one node: the scan summary.
struct NodeSum {
  u64 cost;   // fp64 cost bit pattern
  u32 code;   // scanner slot code (kNone = no node)
  u32 len, type;
  int fcpu;   // front (t = now) summary, as in KParams::f_cpu / f_mem / f_cnt
  u32 fmem;
  u64 fcnt;
};

// UpdateNodeSelectorWithScheduledJob (h:636-642) for a job committed by the inline path in a cycle with preemption (lane 0 of the
// worker; what commit_selection does for the general path): the job joins its node's qos_job_map.  Out of line: the inline path of
// every other cycle must not carry its registers.
__device__ __noinline__ void pre_join_single(const KParams& P, u32 q, u64 poff, u32 orig, i64 end) {
  const u32 hq = P.slot_block ? P.slot_block[q] : q;
  P.pre.rec_orig[poff] = orig; P.pre.rec_slot[poff] = q; P.pre.rec_gone[poff] = 0;
  P.pre.rec_next[poff] = P.pre.slot_head[hq];
  P.pre.slot_head[hq] = (u32)poff;
  P.pre.pj_rec0[orig] = (u32)poff; P.pre.pj_k[orig] = 1; P.pre.pj_end[orig] = end;
}
// The owner updates of a job the inline path committed on a node that several partitions list (one time map per node, a cost per
// partition: JobScheduler.cpp:6563,6609-6617): the own record, then one "keep cost" record per other slot of the node (new length and
// front summary), through the HBM list — where the scanners of a group of partitions that share nodes look — as commit_selection
// writes them.  Lane 0 of the worker; out of line: every other cycle's inline path must not carry its registers.
__device__ __noinline__ void commit_single_shared(const KParams& P, u32 q, u32 qbeg, UpdRec u, int* s_nupd) {
  UpdRec* const upd = P.g_upd + qbeg;
  upd[0] = u;
  u32 nup = 1;
  for (u32 a = P.sib_off[q]; a < P.sib_off[q + 1]; ++a) {
    const u32 qs = P.sib[a];
    UpdRec us = u;
    const u32 ps = qs - qbeg;
    us.p = ((ps / kScan) << 10) | (ps % kScan);
    us.has_front = u.has_front | 2u;
    if (u.has_front) { P.f_cpu[qs] = u.fcpu; P.f_mem[qs] = u.fmem; P.f_cnt[qs] = u.fcnt; }
    if (P.f_len) P.f_len[qs] = u.len;
    upd[nup++] = us;
  }
  *s_nupd = (int)nup;
}
__device__ __forceinline__ void commit_single_regs(const KParams& P, i64 L, u32 orig, u64 poff, NodeHdr* hd,
                                                   const NodeHdr& h, const TlEntry& e, u32 q, u32 code, double cost,
                                                   const Res& alloc, i64 start, int reason, u32 lane, UpdRec* s_upd,
                                                   int* s_nupd, NodeSum& ns, const KParams& Pmem, u32 qbeg) {   // (Pmem: the block's copy in HBM)
  const i64 end = start + L;
  const u32 newlen = tl_commit_regs(P, hd, tl_of(P, hd), e, h.len, start, end, alloc, lane, orig);
  const double ratio = ((double)alloc.cpu / 256.0) / ((double)h.total.cpu / 256.0);
  const double delta = (double)(end - start) * ratio;
  const double ncost = cost + delta;
  const bool has_front = start == P.now;
  Res f = rl_res(e.r, 0);  // entry 0 = the entry at `now`
  if (has_front) res_sub(f, alloc);
  ns.cost = cost_key(ncost); ns.code = code; ns.len = newlen; ns.type = h.type;
  ns.fcpu = clamp_cpu(f.cpu); ns.fmem = mem_mib_ceil(f.mem); ns.fcnt = class_counts(f.gres, P.gres);
  if (lane == 0) {
    UpdRec u;
    u.p = code;
    u.len = newlen;
    u.cost = ncost;
    u.has_front = has_front ? 1u : 0u;
    u.fcpu = ns.fcpu;
    u.fmem = ns.fmem;
    u.fcnt = ns.fcnt;
    u.pad = 0;
    P.cost[q] = ncost;
    if (has_front) { P.f_cpu[q] = u.fcpu; P.f_mem[q] = u.fmem; P.f_cnt[q] = u.fcnt; }
    if (P.sib_off) commit_single_shared(Pmem, q, qbeg, u, s_nupd);
    else { s_upd[0] = u; *s_nupd = 1; }
    P.o_node[poff] = h.node;
    P.o_ntasks[poff] = 1;
    P.o_cpu[poff] = alloc.cpu;
    P.o_mem[poff] = alloc.mem;
    P.o_clo[poff] = alloc.clo;
    P.o_chi[poff] = alloc.chi;
    P.o_gres[poff] = alloc.gres;
    if (P.o_c2) { P.o_c2[poff] = alloc.c2; P.o_c3[poff] = alloc.c3; }
    P.o_start[orig] = start;
    P.o_reason[orig] = (uint8_t)reason;
    if (P.pre.enabled) pre_join_single(Pmem, q, poff, orig, end);
  }
  if (P.sib_off) __threadfence_block();   // (the records went through HBM)
}

// Would node `ns` be a candidate of job X?  b: may host it at all, a: may start it now.  Same meaning as the
// scanners' bmask / amask (any NECESSARY condition is valid for `a`: the exact test decides).
__device__ __forceinline__ void eval_node(const KParams& P, const Req& mv, u32 flags, u64 typeok, const NodeSum& ns,
                                          bool& b, bool& a) {
  b = ns.code != kNone && !req_impossible(mv) && ((typeok >> ns.type) & 1ull) != 0 && ns.len < P.max_jobs_per_node;
  a = b && mv.cpu <= (i64)ns.fcpu && (mv.mem >> 20) <= (u64)ns.fmem;
  if (a && (flags & kJfGres)) {
    const u64 H8 = 0x8080808080808080ull;
    a = (((ns.fcnt | H8) - mv.gspec) & H8) == H8;
    for (int g = 0; g < kMaxNames; ++g) {
      const u32 tot = (mv.gtot >> (8 * g)) & 0xFFu;
      if (tot && byte_sum(ns.fcnt & P.gres.name_bytes[g]) < tot) a = false;
    }
  }
}

// Task distribution over the k selected nodes, smallest capacity first (JobScheduler.cpp:6304-6325 /
// :6345-6367), then the per-node allocation cut out of ent.res.  Leaves H[i].ntasks = tasks on the node
// and H[i].res = allocated_res on the node.  Returns false on an invariant violation.
__device__ __noinline__ bool distribute_and_alloc(const KParams& P, const JobCtx& J, HeapEnt* H, u32 lane) {
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
// k_select — one persistent, wave-specialised workgroup per partition.
// Both roles run the same barrier schedule; they exchange only the per-wave argmin slots, one flag and
// the owner-update records through LDS.  Separate branches => separate register allocation: the tile
// registers are not live in the worker's code and vice versa.
//
// Barrier schedule of one job (W = worker, S = scanners):
//   S: publish A (can-start-now argmin) and T (fits-res_total argmin)                   B1
//   while A != none:   W: exact test [+ commit]  -> verdict                            B2
//                      verdict 2 -> done | else S: next A argmin                       B1
//   A == none  -> phase B: cur = T;  while cur != none and < k nodes: S: next T argmin  B1 (+B2 if ntasks>k)
//                 W: allocations vs res_total, earliest start, commit -> verdict       B3
// ---------------------------------------------------------------------------------------------
struct AllocCacheEnt {  // (request shape, node type) -> allocation against res_total
  i64 cpu; u64 gspec; u32 gtot; u32 type; u64 clo, chi, gres, c2, c3;
};
constexpr int kAllocCache = 256;
struct WorkerShared {
  u64 (*wc)[kRed];
  u32 (*wp)[kRed];
  int* flag;
  int* nupd;
  UpdRec* upd;
  HeapEnt* heap;
  u32 part;   // engine partition this workgroup serves (k_select / k_pipe: blockIdx.x; k_wide: several workgroups per partition)
};

#ifdef CNS_PROF_DIP   // (a -DCNS_PROF -DCNS_PROF_DIP build counts k_select's dips in the slots of the multi-node protocols)
#define PROF_DIP(slot) do { if (lane == 0) P.prof[(size_t)(P.part_map ? P.part_map[blockIdx.x] : blockIdx.x) * 32 + (slot)] += 1; } while (0)
#else
#define PROF_DIP(slot) do {} while (0)
#endif
#ifdef CNS_PROF
#define PROF_T(var) const long long var = clock64()
#define PROF_ADD(slot, a, b) do { if (lane == 0) P.prof[(size_t)(P.part_map ? P.part_map[blockIdx.x] : blockIdx.x) * 32 + (slot)] += (u64)((b) - (a)); } while (0)
#define PROF_CNT(slot) do { if (lane == 0) P.prof[(size_t)(P.part_map ? P.part_map[blockIdx.x] : blockIdx.x) * 32 + (slot)] += 1; } while (0)
#define PROF_ADDS(slot, a, b) do { if (lane == 0 && wave == 1) P.prof[(size_t)(P.part_map ? P.part_map[blockIdx.x] : blockIdx.x) * 32 + (slot)] += (u64)((b) - (a)); } while (0)
#else
#define PROF_T(var)
#define PROF_ADD(slot, a, b)
#define PROF_CNT(slot)
#define PROF_ADDS(slot, a, b)
#endif

#include "preempt_dev.inc"

// k_select's scanners keep one dip per node in their tile (NPL <= kSelDipMaxNpl), like k_wide's (KParams::dip_*): the worker posts
// what a rejected start-now candidate revealed in LDS next to its verdict (fl[4..7]); the node's owner lane takes it after B2 and
// stops proposing the node to jobs whose windows reach that far and which do not fit it (a second NECESSARY condition for
// :6274-6285; the candidates still come in cost order, fewer of them).  Posted is
//   - the first FUTURE entry inside the job's window that by itself cannot host the minimum view (cpu, memory, GRES counts), or
//   - when every entry can: the WINDOW MINIMUM m itself (res_avail folded with every entry of the window: the GRES slots that are
//     free THROUGHOUT — on a loaded cluster the running jobs and the backfilled ones hold different slots at different times),
//     dated at the last entry inside the window: a window that reaches that entry contains all of this one's entries, and entries
//     only shrink and split within a cycle, so its minimum lies below m in every component.
// A release by TryPreempt_ is the one thing that invalidates a dip: the owner updates of a preempting job carry kUpdReleased and
// the owner forgets the row's dip.  Register maps (<= 64 entries) only.
constexpr int kSelDipMaxNpl = 19;
constexpr u32 kUpdReleased = 4u;   // UpdRec::has_front bit 2
// A row's dip time, cpus and memory in ONE register, every field rounded towards "no effect": 16 s units rounded UP (a window
// in 16 s units rounded DOWN that lies beyond it really contains the entry; 0xFFFF: none / more than 12 days away) | whole cpus,
// rounded up by dip_cm_of, capped at 255 | GiB likewise (a request is capped the same way before it is compared: a capped dip fits all).
constexpr u32 kDipNone = 0xFFFF0000u;
__device__ __forceinline__ u32 pack_dip(u32 dt, u32 dcm) {
  const u32 t16 = dt >= 0xFFFF0u ? 0xFFFFu : ((dt + 15u) >> 4);
  const u32 c = dcm >> 16, m = dcm & 0xFFFFu;
  return (t16 << 16) | ((c > 255u ? 255u : c) << 8) | (m > 255u ? 255u : m);
}
__device__ __forceinline__ bool post_dip(const KParams& P, const GresDev& G, int* fl, u32 code, const TlEntry& e, u32 len, const Req& mv, i64 E, u32 lane, const Res& m) {
  if (len > 64) { if (lane == 0) fl[4] = (int)kNone; return false; }
  const bool inw = lane == 0 || (lane < len && e.t < E);
  const bool cand = inw && lane >= 1 && !feasible_counts(mv, e.r.cpu, e.r.mem, 0u, class_counts(e.r.gres, G), G);
  const u64 b = __ballot(cand);
  const u32 i = b ? (u32)__builtin_ctzll(b) : 63u - (u32)__builtin_clzll(__ballot(inw));   // the first entry that fails | the last one inside the window
  const Res r = b ? rl_res(e.r, i) : m;
  const i64 t = (i64)rl64((u64)e.t, i);
  const i64 off = t <= P.now ? 0 : t - P.now;
  if (off >= 0xFFFFFFFFll) { if (lane == 0) fl[4] = (int)kNone; return false; }
  if (lane == 0) { fl[4] = (int)code; fl[5] = (int)(u32)off; fl[6] = (int)dip_cm_of(r); fl[7] = (int)nibbles_of(class_counts(r.gres, G)); }
  return b != 0;
}

// Out-of-line worker path for everything that is not "node_num == 1, ntasks == 1, shared node":
// multi-node jobs, ntasks > node_num (priority_queue emulation) and exclusive jobs.  Enters after the
// round-0 barrier with the A and T winners, leaves after the job's last barrier; returns the LDS
// double-buffer parity.
template <u32 kS = kScan>
__device__ __noinline__ int worker_job_slow(const KParams& Pm, const WorkerShared sh, const JobCtx* Jp, int par,
                                            u64 wc, u32 wcode, u64 tc, u32 tcode, u32 qbeg, HeapEnt* gheap) {
  const auto& P = kparams_scalar(&Pm);   // (Pm: the block's copy in HBM — what the out-of-line routines and the GRES tables are given)
  const JobCtx J = *Jp;
  const u32 lane = threadIdx.x & 63u;
  drain_stores();
  const bool excl_job = (J.flags & kJfExclusive) != 0;
  // ntasks_on_node_total per node type (JobScheduler.cpp:6222): lane t evaluates type t
  int tt_lane = 0;
  if (lane < P.num_types) {
    const Res ttot = P.type_total[lane];
    if (J.general) tt_lane = max_tasks(J.min_view, J.tcpu, J.tmem, J.tmin, J.tmax, ttot, Pm.gres);
    else { Res tmp; tt_lane = feasible(J.min_view, ttot, tmp, Pm.gres) ? (int)J.tmin : 0; }
  }
  const u32 orig = J.orig;
  HeapEnt* const H = (J.k < (u32)kLdsHeap) ? sh.heap : gheap;
  // (a job of the inline path's shape that can preempt nobody, diverted to here by a long time map: the scanners expect its update in LDS)
  const bool via_hbm = J.k > (u32)kMaxUpd || P.sib_off || (P.general_only && !fast_in_preempt_cycle(J.flags, J.k, J.general, J.tmin == 1));
  UpdRec* const s_upd = !via_hbm ? sh.upd : P.g_upd + qbeg;  // long lists (sibling slots of shared nodes, releases of a preemption) go through HBM
  int* const s_nupd = sh.nupd;

  // ---- Phase A: start now (GetNodesAndTrySchedule_, JobScheduler.cpp:6188-6333) -------------
  int hsize = 0, hsum = 0;  // topk_nodes_avail.size(), topk_ntasks_sum_avail
  while (wcode != kNone) {
    const u32 q = qbeg + slot_of_code_t<kS>(wcode);
    NodeHdr* const hd = hdr_of(P, q);
    const TlMap T = tl_of(P, hd);
    int code = 0;
    const u32 len = hd->len;
    const u32 n = hd->node;
    bool ok = false;
    Res m = res_zero();
    int ta = 0;
    if (!excl_job) {
      const Res a0 = hd->avail0;
      Res f;
      if (feasible(J.min_view, a0, f, Pm.gres)) {             // :6274
        m = window_min(T, len, a0, J.E, lane);               // :6278-6283
        if (J.general) ta = max_tasks(J.min_view, J.tcpu, J.tmem, J.tmin, J.tmax, m, Pm.gres);  // :6285
        else ta = feasible(J.min_view, m, f, Pm.gres) ? (int)J.tmin : 0;
        ok = ta > 0;
      }
    } else {
      m = hd->total;
      ok = window_all_total(T, len, m, J.E, lane);          // :6250-6260
      ta = __shfl(tt_lane, (int)hd->type);
    }
    if (ok) {
      HeapEnt x;
      x.ntasks = ta; x.p = wcode; x.node = n; x.pad = 0;
      x.cost = P.cost[q];   // (= the scanners' value: every earlier commit is in; the key may be in signed form, cost_key_m)
      x.res = m;
      int nsum = hsum + ta, nsize = hsize + 1;
      if (lane == 0) {
        H[hsize] = x;
        pq_push(H, nsize);                                                // :6288-6289
        if (nsize > (int)J.k) { nsum -= H[0].ntasks; pq_pop(H, nsize); }  // :6290-6293
      }
      nsum = __shfl(nsum, 0);
      if (nsize > (int)J.k) --nsize;
      hsum = nsum; hsize = nsize;
      __threadfence_block();
      code = 1;
      if (hsize == (int)J.k && (u32)hsum >= J.ntasks) {                   // :6294-6297
        if (!distribute_and_alloc(Pm, J, H, lane)) { if (lane == 0) set_fault(P, 2, orig, n, 2); }
        commit_selection<kS>(Pm, J, H, qbeg, P.now, lane, s_upd, s_nupd);      // start_time = now (:6326)
        if (lane == 0) { P.o_start[orig] = P.now; P.o_reason[orig] = 0; }
        code = 2;
      }
    }
    if constexpr (kS == kScan) {   // (k_select's scanners alone read the dip words: post_dip)
      if (!ok && !excl_job && len <= 64) {
        TlEntry e;
        e.t = kInf;
        if (lane < len) e = T[lane];
        Res f;
        if (feasible(J.min_view, hd->avail0, f, Pm.gres)) post_dip(P, Pm.gres, sh.flag, wcode, e, len, J.min_view, J.E, lane, m);   // (m: the window minimum)
        else post_dip(P, Pm.gres, sh.flag, wcode, e, 1u, J.min_view, J.E, lane, hd->avail0);   // res_avail itself cannot host it (:6274): whatever the window
      } else if (lane == 0) {
        sh.flag[4] = (int)kNone;
      }
    }
    if (via_hbm) __threadfence_block();  // the owner updates went through HBM (g_upd)
    if (lane == 0) *sh.flag = code;
    wg_barrier();  // B2: verdict (and, on success, the owner updates) visible to the scanners
    if (code == 2) return par;
    wg_barrier();  // B1 of the next round
    wc = sh.wc[par][lane & (kRed - 1)];
    wcode = sh.wp[par][lane & (kRed - 1)];
    reduce16(wc, wcode);
    par ^= 1;
    wc = uni64(wc); wcode = uni32(wcode);
  }

  // ---- Phase B: top-k nodes by total capacity, then backfill -----------------------------------
  // (JobScheduler.cpp:6233-6242, :6335-6368, Backfill_ :6371-6376)
  int code = 0;
  int nsel = 0, tsum = 0;
  bool complete = false;
  u64 cc = tc;
  u32 ccode = tcode;
  while (ccode != kNone) {
    NodeHdr* const hd = hdr_of(P, qbeg + slot_of_code_t<kS>(ccode));
    HeapEnt x;
    x.p = ccode; x.node = hd->node; x.pad = 0;
    x.cost = P.cost[qbeg + slot_of_code_t<kS>(ccode)];
    x.res = res_zero();
    if (!J.general) {
      x.ntasks = 1;
      if (lane == 0) H[nsel] = x;
      ++nsel;
      if (nsel == (int)J.k) { complete = true; break; }
    } else {
      const int tt = __shfl(tt_lane, (int)hd->type);
      x.ntasks = tt;
      int nsum = tsum + tt, nsize = nsel + 1;  // the push condition (:6233-6234) held, else we had stopped
      if (lane == 0) {
        H[nsel] = x;
        pq_push(H, nsize);
        if (nsize > (int)J.k) { nsum -= H[0].ntasks; pq_pop(H, nsize); }
      }
      nsum = __shfl(nsum, 0);
      if (nsize > (int)J.k) --nsize;
      tsum = nsum; nsel = nsize;
      const bool stop = nsel == (int)J.k && (u32)tsum >= J.ntasks;
      if (lane == 0) *sh.flag = stop ? 1 : 0;
      wg_barrier();  // B2 (ntasks > node_num only)
      if (stop) { complete = true; break; }
    }
    wg_barrier();  // B1
    cc = sh.wc[par][lane & (kRed - 1)];
    ccode = sh.wp[par][lane & (kRed - 1)];
    reduce16(cc, ccode);
    par ^= 1;
    cc = uni64(cc); ccode = uni32(ccode);
  }
  if (complete) {
    __threadfence_block();
    for (u32 i = lane; i < J.k; i += 64) {
      HeapEnt x = H[i];
      x.res = hdr_of(P, qbeg + slot_of_code_t<kS>(x.p))->total;
      H[i] = x;
      P.bf_j[qbeg + i] = 0;
    }
    __threadfence_block();
    if (!distribute_and_alloc(Pm, J, H, lane)) { if (lane == 0) set_fault(P, 3, orig, 0, 1); }
    // TryPreempt_ before Backfill_ (JobScheduler.cpp:6140-6143), only in a cycle that was started with preemption
    bool preempted = false;
    if (P.pre.enabled) {
      u32 pf = 0;
      int nch = -1;
      // (only k_select runs cycles with preemption, and only its instantiation reaches g_pre_cache: the 100 KB of LDS are allocated
      // in k_select's kernels alone — a pointer through WorkerShared cost every kernel two more argument registers at each call of
      // an out-of-line worker routine, and k_select its spill-free allocation)
      if constexpr (kS == kScan) nch = pre_try<kS>(Pm, J, H, qbeg, sh.part, &pf);
      else pf = 33;
      if (pf && lane == 0) set_fault(P, pf, orig, J.k, 0);
      if (nch >= 0) {
        const u32 nn = P.part_off[sh.part + 1] - qbeg;
        u32* const touched = P.bf_j + qbeg;   // (the backfill cursors are not needed on this branch)
        const u32 nt = pre_release(Pm, J, qbeg, nn, sh.part, (u32)nch, touched);
        for (u32 i = lane; i < J.k; i += 64) H[i].cost = P.cost[qbeg + slot_of_code_t<kS>(H[i].p)];   // the releases lowered costs
        __threadfence_block();
        commit_selection<kS>(Pm, J, H, qbeg, P.now, lane, s_upd, s_nupd);   // start_time = now (:6501), h:630-634
        // the scanners' rows of the nodes that were released on (other than the job's own: their records are final)
        u32 nup = (u32)*s_nupd;
        if (lane == 0)   // every record of this job tells the owner that resources came BACK: a dip it holds for the row is void
          for (u32 i = 0; i < nup; ++i) s_upd[i].has_front |= kUpdReleased;
        for (u32 x = 0; x < nt; ++x) {
          const u32 q = touched[x];
          bool dup = false;
          for (u32 i = 0; i < J.k; ++i) dup = dup || qbeg + slot_of_code_t<kS>(H[i].p) == q;
          if (dup) continue;
          NodeHdr* hd = hdr_of(P, q);
          if (lane == 0) {
            UpdRec u;
            const u32 ps = q - qbeg;
            u.p = ((ps / kS) << 10) | (ps % kS);
            u.len = hd->len; u.cost = P.cost[q]; u.fcpu = P.f_cpu[q]; u.fmem = P.f_mem[q]; u.fcnt = P.f_cnt[q];
            u.has_front = 1u | kUpdReleased; u.pad = 0;
            s_upd[nup] = u;
          }
          ++nup;
          if (P.sib_off)   // the released node's slots in the other partitions of the group (new length / front summary, own cost kept)
            for (u32 a = P.sib_off[q]; a < P.sib_off[q + 1]; ++a) {
              if (lane == 0) {
                UpdRec us;
                const u32 ps = P.sib[a] - qbeg;
                us.p = ((ps / kS) << 10) | (ps % kS);
                us.len = hd->len; us.cost = 0.0; us.fcpu = P.f_cpu[q]; us.fmem = P.f_mem[q]; us.fcnt = P.f_cnt[q];
                us.has_front = 3u | kUpdReleased; us.pad = 0;
                s_upd[nup] = us;
              }
              ++nup;
            }
        }
        if (lane == 0) { *s_nupd = (int)nup; P.o_start[orig] = P.now; P.o_reason[orig] = 0; }
        code = 2;
        preempted = true;
      }
    }
    // EarliestStartSubsetSelector::CalcEarliestStartTime as a fixed point over the k nodes
    i64 t = P.now;
    bool found = false;
    for (u32 iter = 0; iter < (1u << 22) && !preempted; ++iter) {
      i64 Tm = t;
      for (u32 i = 0; i < J.k; ++i) {
        const HeapEnt x = H[i];
        NodeHdr* hd = hdr_of(P, qbeg + slot_of_code_t<kS>(x.p));
        const i64 sx = next_fit_wave(tl_of(P, hd), hd->len, &H[i].res, J.L, t);
        Tm = sx > Tm ? sx : Tm;
      }
      if (Tm == kInf || Tm - P.now > P.max_window) break;  // kAlgoMaxTimeWindow, JobScheduler.h:815
      if (Tm == t) { found = true; break; }
      t = Tm;
    }
    if (found) {
      int reason = 0;
      if (t != P.now) {  // JobScheduler.cpp:6797-6833
        bool notle = false, reserved = false;
        for (u32 i = lane; i < J.k; i += 64) {
          const HeapEnt x = H[i];
          const u32 qx = qbeg + slot_of_code_t<kS>(x.p);
          if (!res_le(x.res, hdr_of(P, qx)->avail0)) notle = true;
          if (P.first_resv[qx] < P.now + J.L) reserved = true;
        }
        const bool resv_part = sh.part >= P.num_real_parts;
        reason = (!resv_part && __any(reserved)) ? 3 /*Resource Reserved*/ : (__any(notle) ? 2 /*Resource*/ : 1 /*Priority*/);
      }
      commit_selection<kS>(Pm, J, H, qbeg, t, lane, s_upd, s_nupd);
      if (lane == 0) { P.o_start[orig] = t; P.o_reason[orig] = (uint8_t)reason; }
      code = 2;
    }
  }
  if (code == 0 && lane == 0) { P.o_start[orig] = 0; P.o_reason[orig] = 2; }  // "Resource", :6768
  if (via_hbm) __threadfence_block();  // the owner updates went through HBM (g_upd)
  if (lane == 0) *sh.flag = code;
  wg_barrier();  // B3
  return par;
}

// kW = false instantiations of the tester / commit functions of k_pipe and k_wide serve snapshots WITHOUT core ids above 127
// (KParams::wide_cores == 0, chosen once per call by a scalar branch): every Res that enters them from memory has its upper
// two core words replaced by the constant 0, so the compiler folds their arithmetic, their registers and their loads away —
// those clusters run the code they ran before ABI 3 (the 4-word Res cost the testers of C4 +26 % per test otherwise).
template <bool kW> __device__ __forceinline__ Res narrow(Res r) { if (!kW) { r.c2 = 0; r.c3 = 0; } return r; }
// Loads the node block of slot q the way the fast paths want it: header scalarised, lane i <- entry i.
// kDrain = false: the caller has drained this wave's stores itself and has OTHER loads in flight that the block's loads may overlap
// (k_wide's tester: the job record) — the initial s_waitcnt vmcnt(0) would serialise the two round trips.
template <bool kW = true, bool kDrain = true>
__device__ __forceinline__ void load_block(const KParams& P, u32 q, u32 lane, NodeHdr*& hd, NodeHdr& h, TlEntry& e) {
  if (kDrain) drain_stores();
  hd = hdr_of(P, q);
  if (kW) {
    h = *hd;
    e = tl_of(P, hd)[lane];
  } else {   // the header's first 112 bytes as seven 16-byte loads (one cache line), the 48-byte record as three
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const auto* hq = as_global((const u64x2*)hd);
    const auto* vq = as_global((const u64x2*)(tl_of<false>(P, hd).m + lane));
    asm volatile("" : "+v"(hq), "+v"(vq));   // (both addresses exist before the first load is issued: no address arithmetic between the loads)
    const u64x2 q0 = hq[0], q1 = hq[1], q2 = hq[2], q3 = hq[3], q4 = hq[4], q5 = hq[5], q6 = hq[6];
    h.len = (u32)q0.x; h.node = (u32)(q0.x >> 32); h.type = (u32)q0.y; h.pad = 0;
    h.avail0.cpu = (i64)q1.x; h.avail0.mem = q1.y; h.avail0.clo = q2.x; h.avail0.chi = q2.y; h.avail0.gres = q3.x;
    h.total.cpu = (i64)q4.y; h.total.mem = q5.x; h.total.clo = q5.y; h.total.chi = q6.x; h.total.gres = q6.y;
    h.avail0.c2 = 0; h.avail0.c3 = 0; h.total.c2 = 0; h.total.c3 = 0;
    const u64x2 v0 = vq[0], v1 = vq[1], v2 = vq[2];
    e.t = (i64)v0.x; e.r.cpu = (i64)v0.y; e.r.mem = v1.x; e.r.clo = v1.y; e.r.chi = v2.x; e.r.gres = v2.y; e.r.c2 = 0; e.r.c3 = 0;
  }
  h.len = uni32(h.len); h.node = uni32(h.node); h.type = uni32(h.type);
  h.avail0 = uni_res(h.avail0); h.total = uni_res(h.total);
}

// The exact test rejected this slot for a job that the front (the entry at `now`) admits: a FUTURE entry inside the job's window
// holds less.  The first such entry that by itself cannot host the job (cpu, memory, GRES counts) becomes the slot's dip
// (KParams::dip_*): when the scanners reload their tile they stop proposing the node to jobs whose windows reach it.  Written
// where the misprediction is found — nothing on the path of a prediction that holds.  Register maps (<= 64 entries) only.
// (G: the GRES layout — P.gres, or the caller's own copy of it when P is a scalar copy of the block: kparams_scalar)
__device__ __forceinline__ void record_dip(const KParams& P, const GresDev& G, u32 q, const TlEntry& e, u32 len, const Req& mv, i64 E, u32 lane) {
  if (len > 64) return;
  const bool cand = lane >= 1 && lane < len && e.t < E && e.t - P.now < 0xFFFFFFFFll &&
                    !feasible_counts(mv, e.r.cpu, e.r.mem, 0u, class_counts(e.r.gres, G), G);
  const u64 b = __ballot(cand);
  if (!b) return;
  const u32 i = (u32)__builtin_ctzll(b);
  const Res r = rl_res(e.r, i);
  const i64 t = (i64)rl64((u64)e.t, i);
  if (lane == 0) { P.dip_t[q] = (u32)(t - P.now); P.dip_cm[q] = dip_cm_of(r); P.dip_g[q] = nibbles_of(class_counts(r.gres, G)); }
}
__device__ __forceinline__ void record_dip(const KParams& P, u32 q, const TlEntry& e, u32 len, const Req& mv, i64 E, u32 lane) {
  record_dip(P, P.gres, q, e, len, mv, E, lane);
}

// Commit pick i of a multi-node selection (time map, cost, owner update i); H[i].res = its allocation.
__device__ __forceinline__ void commit_pick(const KParams& P, const JobCtx& J, const HeapEnt& x, u32 i, u32 qbeg,
                                            i64 start, u32 lane, UpdRec* s_upd, u32 orig) {
  const u32 q = qbeg + slot_of_code(x.p);
  NodeHdr* hd; NodeHdr h; TlEntry e;
  load_block(P, q, lane, hd, h, e);
  const i64 end = start + J.L;
  const Res e0 = rl_res(e.r, 0);  // entry at `now`
  u32 newlen;
  if (h.len <= 64) newlen = tl_commit_regs(P, hd, tl_of(P, hd), e, h.len, start, end, x.res, lane, orig);
  else newlen = tl_commit(P, hd, start, end, x.res, lane, orig);
  const double ratio = ((double)x.res.cpu / 256.0) / ((double)h.total.cpu / 256.0);
  const double ncost = x.cost + (double)(end - start) * ratio;
  if (lane == 0) {
    UpdRec u;
    u.p = x.p; u.len = newlen; u.cost = ncost;
    u.has_front = (start == P.now) ? 1u : 0u;
    Res f = e0;
    if (u.has_front) res_sub(f, x.res);
    u.fcpu = clamp_cpu(f.cpu); u.fmem = mem_mib_ceil(f.mem); u.fcnt = class_counts(f.gres, P.gres); u.pad = 0;
    P.cost[q] = ncost;
    if (u.has_front) { P.f_cpu[q] = u.fcpu; P.f_mem[q] = u.fmem; P.f_cnt[q] = u.fcnt; }
    s_upd[i] = u;
  }
}

// placement records of a k-node selection, ascending node index (k <= 64: lane i holds pick i)
__device__ __forceinline__ void emit_placements(const KParams& P, const JobCtx& J, const HeapEnt* H, u32 lane) {
  const u64 poff = J.poff;
  if (lane < J.k) {
    const HeapEnt me = H[lane];
    u32 rank = 0;
    for (u32 m = 0; m < J.k; ++m) rank += H[m].node < me.node ? 1u : 0u;
    const u64 o = poff + rank;
    P.o_node[o] = me.node; P.o_ntasks[o] = 1;
    P.o_cpu[o] = me.res.cpu; P.o_mem[o] = me.res.mem; P.o_clo[o] = me.res.clo; P.o_chi[o] = me.res.chi;
    P.o_gres[o] = me.res.gres;
    if (P.o_c2) { P.o_c2[o] = me.res.c2; P.o_c3[o] = me.res.c3; }
  }
}

// UpdateNodeSelectorWithScheduledJob (h:636-642) for a multi-node job committed by the parallel protocol or by worker_job_multi in a
// cycle with preemption (what commit_selection does for the general path): the job joins the qos_job_map of each of its k nodes, under
// the placement record it wrote there (poff + rank by node index).  One wave (the worker); H[i].node as the helpers / the commit left it.
__device__ __noinline__ void pre_join_multi(const KParams& P, const JobCtx& J, const HeapEnt* H, u32 qbeg, i64 end) {
  const u32 lane = threadIdx.x & 63u;
  if (lane < J.k) {
    const HeapEnt me = H[lane];
    u32 rank = 0;
    for (u32 m = 0; m < J.k; ++m) rank += H[m].node < me.node ? 1u : 0u;
    const u64 o = J.poff + rank;
    const u32 q = qbeg + slot_of_code(me.p);
    P.pre.rec_orig[o] = J.orig; P.pre.rec_slot[o] = q; P.pre.rec_gone[o] = 0;
    const u32 hq = P.slot_block ? P.slot_block[q] : q;
    P.pre.rec_next[o] = P.pre.slot_head[hq];   // (the k nodes are distinct: no two lanes touch one list)
    P.pre.slot_head[hq] = (u32)o;
  }
  if (lane == 0) { P.pre.pj_rec0[J.orig] = (u32)J.poff; P.pre.pj_k[J.orig] = J.k; P.pre.pj_end[J.orig] = end; }
}

// Out-of-line worker path for multi-node jobs with ntasks == node_num on shared nodes (2 <= k <= kMaxUpd):
// same barrier schedule as the general path, but every node is handled with the one-read block primitives.
__device__ __noinline__ int worker_job_multi(const KParams& P, const WorkerShared sh, const JobCtx* Jp, int par,
                                             u64 wc, u32 wcode, u64 tc, u32 tcode, u32 qbeg) {
  const JobCtx J = *Jp;
  const u32 lane = threadIdx.x & 63u;
  const u32 orig = J.orig;
  HeapEnt* const H = sh.heap;
  const Req one_view = compose(J.node_view, J.tcpu, J.tmem, 1);
  const u64 smode = P.general_only ? ~0ull : 0ull;   // (a cycle with preemption: the scanners' keys are in signed form, cost_key_m)

  // ---- Phase A: the first k nodes in cost order that can start the job now (:6188-6333) ----------
  u32 npick = 0;
  while (wcode != kNone) {
    const u32 q = qbeg + slot_of_code(wcode);
    NodeHdr* hd; NodeHdr h; TlEntry e;
    load_block(P, q, lane, hd, h, e);
    int code = 0;
    Res f, m;
    bool ok = false;
    if (feasible(J.min_view, h.avail0, f, P.gres)) {
      m = uni_res(h.len <= 64 ? window_min_regs(e, lane < h.len, h.avail0, J.E)
                              : window_min(tl_of(P, hd), h.len, h.avail0, J.E, lane));
      ok = feasible(J.min_view, m, f, P.gres);
    }
    if (ok) {
      Res alloc = f;
      if (J.tmin != 1 && !feasible(one_view, m, alloc, P.gres)) { if (lane == 0) set_fault(P, 2, orig, h.node, 3); }
      if (lane == 0) {
        HeapEnt x; x.ntasks = 1; x.p = wcode; x.node = h.node; x.pad = 0;
        x.cost = cost_of_key_m(wc, smode); x.res = alloc;
        H[npick] = x;
      }
      ++npick;
      code = 1;
      if (npick == J.k) {  // :6294-6297
        __threadfence_block();
        for (u32 i = 0; i < J.k; ++i) commit_pick(P, J, H[i], i, qbeg, P.now, lane, sh.upd, orig);
        emit_placements(P, J, H, lane);
        if (P.pre.enabled) pre_join_multi(P, J, H, qbeg, P.now + J.L);
        if (lane == 0) { *sh.nupd = (int)J.k; P.o_start[orig] = P.now; P.o_reason[orig] = 0; }
        code = 2;
      }
    }
    if (!ok && h.len <= 64) {
      if (feasible(J.min_view, h.avail0, f, P.gres)) post_dip(P, P.gres, sh.flag, wcode, e, h.len, J.min_view, J.E, lane, m);
      else post_dip(P, P.gres, sh.flag, wcode, e, 1u, J.min_view, J.E, lane, h.avail0);
    } else if (lane == 0) {
      sh.flag[4] = (int)kNone;
    }
    if (lane == 0) *sh.flag = code;
    wg_barrier();  // B2
    if (code == 2) return par;
    wg_barrier();  // B1 of the next round
    wc = sh.wc[par][lane & (kRed - 1)];
    wcode = sh.wp[par][lane & (kRed - 1)];
    reduce16(wc, wcode);
    par ^= 1;
    wc = uni64(wc); wcode = uni32(wcode);
  }

  // ---- Phase B: first k nodes by res_total in cost order, earliest common start (:6335-6376) -------
  int code = 0;
  u32 nsel = 0;
  bool complete = false;
  u64 cc = tc;
  u32 ccode = tcode;
  while (ccode != kNone) {
    if (lane == 0) {
      HeapEnt x; x.ntasks = 1; x.p = ccode; x.node = 0; x.pad = 0;
      x.cost = cost_of_key_m(cc, smode); x.res = res_zero();
      H[nsel] = x;
    }
    ++nsel;
    if (nsel == J.k) { complete = true; break; }
    wg_barrier();  // B1
    cc = sh.wc[par][lane & (kRed - 1)];
    ccode = sh.wp[par][lane & (kRed - 1)];
    reduce16(cc, ccode);
    par ^= 1;
    cc = uni64(cc); ccode = uni32(ccode);
  }
  if (complete) {
    __threadfence_block();
    bool bad = false, notle = false, reserved = false;
    if (lane < J.k) {  // allocation against res_total (:6353-6361)
      HeapEnt x = H[lane];
      const NodeHdr* hd = hdr_of(P, qbeg + slot_of_code(x.p));
      x.node = hd->node;
      Res a;
      if (!feasible(one_view, hd->total, a, P.gres)) bad = true; else x.res = a;
      notle = !res_le(x.res, hd->avail0);
      reserved = P.first_resv[qbeg + slot_of_code(x.p)] < P.now + J.L;
      H[lane] = x;
    }
    if (__any(bad) && lane == 0) set_fault(P, 3, orig, 0, 2);
    notle = __any(notle);
    reserved = __any(reserved) && sh.part < P.num_real_parts;   // (the partition, not the workgroup index: k_wide and split launches differ)
    __threadfence_block();
    // EarliestStartSubsetSelector::CalcEarliestStartTime as a fixed point over the k nodes
    i64 t = P.now;
    bool found = false;
    for (u32 iter = 0; iter < (1u << 20); ++iter) {
      i64 Tm = t;
      for (u32 i = 0; i < J.k; ++i) {
        const HeapEnt x = H[i];
        NodeHdr* hd; NodeHdr h; TlEntry e;
        load_block(P, qbeg + slot_of_code(x.p), lane, hd, h, e);
        i64 s;
        if (h.len <= 64) s = next_fit_regs(e, h.len, x.res, J.L, t, lane);
        else s = next_fit_wave(tl_of(P, hd), h.len, &x.res, J.L, t);
        Tm = s > Tm ? s : Tm;
      }
      if (Tm == kInf || Tm - P.now > P.max_window) break;  // kAlgoMaxTimeWindow, JobScheduler.h:815
      if (Tm == t) { found = true; break; }
      t = Tm;
    }
    if (found) {
      int reason = 0;
      if (t != P.now) reason = reserved ? 3 /*Resource Reserved*/ : (notle ? 2 /*Resource*/ : 1 /*Priority*/);  // :6797-6831
      for (u32 i = 0; i < J.k; ++i) commit_pick(P, J, H[i], i, qbeg, t, lane, sh.upd, orig);
      emit_placements(P, J, H, lane);
      if (P.pre.enabled) pre_join_multi(P, J, H, qbeg, t + J.L);
      if (lane == 0) { *sh.nupd = (int)J.k; P.o_start[orig] = t; P.o_reason[orig] = (uint8_t)reason; }
      code = 2;
    }
  }
  if (code == 0 && lane == 0) { P.o_start[orig] = 0; P.o_reason[orig] = 2; }  // "Resource", :6768
  if (lane == 0) *sh.flag = code;
  wg_barrier();  // B3
  return par;
}

// ---- multi-node jobs: candidate i of the selection is verified / committed by ONE wave, all candidates in
// parallel on the scanner waves ("helpers").  H[i] carries (cost, slot code) in, (node, allocation, ok) out.
// ---------------------------------------------------------------------------------------------
// Parallel protocol for multi-node jobs (2 <= node_num <= kMultiK, ntasks == node_num, shared nodes).
// The k candidate nodes are listed in H[0..k) (cost order).  ALL 16 waves call these routines (they
// contain workgroup barriers); wave i+1 is the helper of candidate i (`active`), the others only follow
// the barriers.  A helper reads its node block ONCE and keeps it in registers from the exact test to
// the commit.
// ---------------------------------------------------------------------------------------------
// Commit of candidate i from the helper's registers + owner update i + its placement record.
__device__ __forceinline__ void helper_commit_regs(const KParams& P, const GresDev& G, const JobCtx& J, const HeapEnt* H, u32 i, u32 p,
                                                   double cost0, NodeHdr* hd, const NodeHdr& h, const TlEntry& e,
                                                   const Res& res, i64 start, u32 lane, UpdRec* upd, u32 q, u32 sib_at, u32 qbeg) {
  const i64 end = start + J.L;
  const Res e0 = rl_res(e.r, 0);  // entry at `now`
  u32 newlen;
  if (h.len <= 64) newlen = tl_commit_regs(P, hd, tl_of(P, hd), e, h.len, start, end, res, lane, J.orig);
  else newlen = tl_commit(P, hd, start, end, res, lane, J.orig);
  // MinCpuTimeRatioFirst::UpdateCost, JobScheduler.h:47-53 — ratio first, then x seconds, then +=
  const double ratio = ((double)res.cpu / 256.0) / ((double)h.total.cpu / 256.0);
  const double ncost = cost0 + (double)(end - start) * ratio;
  if (lane == 0) {
    UpdRec u;
    u.p = p; u.len = newlen; u.cost = ncost;
    u.has_front = (start == P.now) ? 1u : 0u;
    Res f = e0;
    if (u.has_front) res_sub(f, res);
    u.fcpu = clamp_cpu(f.cpu); u.fmem = mem_mib_ceil(f.mem); u.fcnt = class_counts(f.gres, G); u.pad = 0;
    P.cost[q] = ncost;
    if (u.has_front) { P.f_cpu[q] = u.fcpu; P.f_mem[q] = u.fmem; P.f_cnt[q] = u.fcnt; }
    upd[i] = u;
    if (P.sib_off) {   // the node's slots in the other partitions of the group: "keep cost" records behind the k own ones, from sib_at
      u32 x = sib_at;
      for (u32 a = P.sib_off[q]; a < P.sib_off[q + 1]; ++a) {
        const u32 qs = P.sib[a];
        UpdRec us = u;
        const u32 ps = qs - qbeg;
        us.p = ((ps / kScan) << 10) | (ps % kScan);
        us.has_front = u.has_front | 2u;
        if (u.has_front) { P.f_cpu[qs] = u.fcpu; P.f_mem[qs] = u.fmem; P.f_cnt[qs] = u.fcnt; }
        if (P.f_len) P.f_len[qs] = newlen;
        upd[x++] = us;
      }
    }
    u32 rank = 0;  // placement records are listed by ascending node index
    for (u32 m = 0; m < J.k; ++m) rank += H[m].node < h.node ? 1u : 0u;
    const u64 o = J.poff + rank;
    P.o_node[o] = h.node; P.o_ntasks[o] = 1;
    P.o_cpu[o] = res.cpu; P.o_mem[o] = res.mem; P.o_clo[o] = res.clo; P.o_chi[o] = res.chi; P.o_gres[o] = res.gres;
    if (P.o_c2) { P.o_c2[o] = res.c2; P.o_c3[o] = res.c3; }
  }
}

// Start-now ending (:6188-6333 with the first k nodes in cost order): exact test of every candidate in
// parallel; if all pass, commit them at `now`.  Returns false (nothing written) if any candidate failed.
__device__ __noinline__ bool multi_verify_commit(const KParams& P, const GresDev* Gp, const JobCtx* Jp, HeapEnt* H, u32 i,
                                                 bool active, u32 qbeg, UpdRec* upd, int* nupd) {
  const u32 lane = threadIdx.x & 63u;
  const GresDev& G = *Gp;
  const JobCtx J = *Jp;
  NodeHdr* hd = nullptr;
  NodeHdr h;
  TlEntry e;
  Res f = res_zero();
  u32 p = 0, q = 0;
  double cost0 = 0.0;
  h.len = 0;
  if (active) {
    p = H[i].p;
    cost0 = H[i].cost;
    q = qbeg + slot_of_code(p);
    load_block(P, q, lane, hd, h, e);
    // :6285 first; it implies :6274 except for the core-id count of res_avail (see the single-node fast path)
    const Res m = uni_res(h.len <= 64 ? window_min_regs(e, lane < h.len, h.avail0, J.E)
                                      : window_min(tl_of(P, hd), h.len, h.avail0, J.E, lane));    // :6278-6283
    bool ok = feasible(J.min_view, m, f, G);  // tpn_min == 1: f is the 1-task allocation (:6285, :6312-6320)
    if (ok) {
      const i64 req_int = J.min_view.cpu / 256;
      const u32 nc0 = cores_count(h.avail0);
      if (req_int * 256 == J.min_view.cpu && nc0 != 0 && nc0 < (u32)req_int) ok = false;   // :528-534 on res_avail
    }
    if (lane == 0) { H[i].node = h.node; H[i].ntasks = ok ? 1 : 0; H[i].res = f; H[i].pad = P.sib_off ? (P.sib_off[q + 1] - P.sib_off[q]) << 8 : 0u; }
  }
  wg_barrier();  // M3: verdicts in
  u32 nok = 0;
  for (u32 m = 0; m < J.k; ++m) nok += H[m].ntasks != 0 ? 1u : 0u;
  if (nok != J.k) return false;  // (rare) the caller falls back to the sequential protocol
  // a group of partitions that shares nodes: the records go through the HBM list, the sibling slots of helper m's node behind the k own ones
  u32 sib_at = J.k, total = J.k;
  if (P.sib_off) {
    upd = P.g_upd + qbeg;
    for (u32 m = 0; m < J.k; ++m) { const u32 ns = H[m].pad >> 8; sib_at += m < i ? ns : 0u; total += ns; }
  }
  if (active) {
    helper_commit_regs(P, G, J, H, i, p, cost0, hd, h, e, f, P.now, lane, upd, q, sib_at, qbeg);
    drain_stores();  // the worker reads this block again in later jobs
    if (P.sib_off) __threadfence_block();
  }
  if (threadIdx.x == 0) *nupd = (int)total;
  wg_barrier();  // M4: commits + owner updates visible
  return true;
}

// Backfill ending (:6335-6376): allocations against res_total, earliest common start as the fixed point
// t <- max_i next_fit_i(t) (each helper evaluates its own node, one barrier per iteration), commit at t.
// Returns the start time or kInf (nothing written).  nf: 2 x kMultiK exchange slots in LDS.
__device__ __noinline__ i64 multi_backfill_par(const KParams& P, const GresDev* Gp, const JobCtx* Jp, HeapEnt* H, u32 i,
                                               bool active, u32 qbeg, UpdRec* upd, int* nupd, i64* nf, int* reason_out) {
  const u32 lane = threadIdx.x & 63u;
  const GresDev& G = *Gp;
  const JobCtx J = *Jp;
  NodeHdr* hd = nullptr;
  NodeHdr h;
  TlEntry e;
  Res alloc = res_zero();
  u32 p = 0, q = 0;
  double cost0 = 0.0;
  h.len = 0;
  if (active) {
    p = H[i].p;
    cost0 = H[i].cost;
    q = qbeg + slot_of_code(p);
    load_block(P, q, lane, hd, h, e);
    if (!feasible(J.min_view, h.total, alloc, G)) {  // :6354-6356
      if (lane == 0) set_fault(P, 3, J.orig, h.node, 2);
    }
    if (lane == 0) {
      H[i].node = h.node; H[i].ntasks = 1; H[i].res = alloc;
      H[i].pad = (res_le(alloc, h.avail0) ? 0u : 1u) | (P.first_resv[q] < P.now + J.L ? 2u : 0u) | (P.sib_off ? (P.sib_off[q + 1] - P.sib_off[q]) << 8 : 0u);
    }
  }
  i64 t = P.now;
  bool found = false;
  int par2 = 0;
  for (u32 iter = 0; iter < (1u << 20); ++iter) {
    if (active) {
      const i64 sx = h.len <= 64 ? next_fit_regs(e, h.len, alloc, J.L, t, lane) : next_fit_wave(tl_of(P, hd), h.len, &alloc, J.L, t);
      if (lane == 0) nf[par2 * kMultiK + (int)i] = sx;
    }
    wg_barrier();
    i64 Tm = t;
    for (u32 m = 0; m < J.k; ++m) {
      const i64 sx = nf[par2 * kMultiK + (int)m];
      Tm = sx > Tm ? sx : Tm;
    }
    par2 ^= 1;
    if (Tm == kInf || Tm - P.now > P.max_window) break;  // kAlgoMaxTimeWindow, JobScheduler.h:815
    if (Tm == t) { found = true; break; }
    t = Tm;
  }
  bool notle = false, reserved = false;
  for (u32 m = 0; m < J.k; ++m) { notle = notle || (H[m].pad & 1u) != 0; reserved = reserved || (H[m].pad & 2u) != 0; }
  if (qbeg >= P.part_off[P.num_real_parts]) reserved = false;  // jobs of a reservation (its slots lie behind the real partitions'): no "Resource Reserved" (:6798,6818)
  u32 sib_at = J.k, total = J.k;
  if (P.sib_off) {
    upd = P.g_upd + qbeg;
    for (u32 m = 0; m < J.k; ++m) { const u32 ns = H[m].pad >> 8; sib_at += m < i ? ns : 0u; total += ns; }
  }
  if (found && active) {
    helper_commit_regs(P, G, J, H, i, p, cost0, hd, h, e, alloc, t, lane, upd, q, sib_at, qbeg);
    drain_stores();
    if (P.sib_off) __threadfence_block();
  }
  if (found && threadIdx.x == 0) *nupd = (int)total;
  wg_barrier();  // commits + owner updates visible (or: nothing happened)
  *reason_out = (found && t != P.now) ? (reserved ? 3 : (notle ? 2 : 1)) : 0;  // :6797-6831
  return found ? t : kInf;
}

// ---- scanner tile compression -----------------------------------------------------------------------
// Per node the scanners keep 5 dwords: fp64 cost (exact), front cpu (exact i32) and two packed words
//   mw = front mem in GiB rounded up (16 bit, saturating) | time-map length << 16 | node type << 26
//   gn = front GRES popcount per class, 4 bits each, saturating at 15
// The front values only feed a NECESSARY condition (the worker's exact test decides), so rounding the
// node side up and the request side down keeps the filter conservative.
__device__ __forceinline__ u32 mem_gib16(u32 mib) { u32 g = (mib + 1023u) >> 10; return g > 0xFFFFu ? 0xFFFFu : g; }
__device__ __forceinline__ u32 nibbles_of(u64 cnt) {  // byte g -> nibble g, saturating at 15
  const u64 hi = (cnt >> 4) & 0x0F0F0F0F0F0F0F0Full;
  const u64 nz = ((hi + 0x0F0F0F0F0F0F0F0Full) >> 4) & 0x0101010101010101ull;  // 1 where the byte is >= 16
  u64 x = (cnt & 0x0F0F0F0F0F0F0F0Full) | (nz * 15ull);
  x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
  x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
  x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
  return (u32)x;
}
// inverse direction for the worker's merge: nibble g -> byte g; a saturated nibble (15 = "15 or more")
// becomes 127, the largest count a request byte can hold (req_impossible), so the test stays necessary
__device__ __forceinline__ u64 bytes_of_nibbles(u32 nb) {
  u64 lo = nb & 0xFFFFu, hi = nb >> 16;
  u64 v = lo | (hi << 32);                       // 4 nibbles per 32-bit half
  v = (v | (v << 8)) & 0x00FF00FF00FF00FFull;
  v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0Full;
  const u64 sat = (v + 0x0101010101010101ull) & 0x1010101010101010ull;  // byte == 15 -> bit 4 set
  return v | ((sat >> 4) * 0x70ull);             // 15 -> 127
}

// included / excluded node lists of a job (JobScheduler.cpp:6202-6220) for the nodes of one scanner lane
// whose bit is set in bmask; out of line: rare, and it touches no tile register.
template <u32 kS = kScan>
__device__ __noinline__ u64 list_mask(const KParams* Pp, u32 flags, u64 ji, u64 bmask, u32 slot0, u32 npl) {
  const u32* rec = Pp->jobrec + ji * kJobRecDwords;  // rare path: the list bounds are read from the record itself
  const u64 incl_b = ((u64)rec[kJrInclB + 1] << 32) | rec[kJrInclB], incl_e = ((u64)rec[kJrInclE + 1] << 32) | rec[kJrInclE];
  const u64 excl_b = ((u64)rec[kJrExclB + 1] << 32) | rec[kJrExclB], excl_e = ((u64)rec[kJrExclE + 1] << 32) | rec[kJrExclE];
  u64 lm = 0;
  for (u32 r = 0; r < npl; ++r) {
    if (!((bmask >> r) & 1ull)) continue;
    const u32 n = Pp->slot_node[slot0 + r * kS];
    bool okl = true;
    if ((flags & kJfIncl) && !in_list(Pp->incl_nodes, incl_b, incl_e, n)) okl = false;
    if ((flags & kJfExcl) && in_list(Pp->excl_nodes, excl_b, excl_e, n)) okl = false;
    lm |= (u64)(okl ? 1u : 0u) << r;
  }
  return lm;
}

template <u32 V> struct ModeTag { static constexpr u32 value = V; };
template <bool Wide> struct RowMask { using type = u32; };
template <> struct RowMask<true> { using type = u64; };

template <int NPL>
__global__ __launch_bounds__(kBlock) void k_select(const KParams P, const KParams* __restrict__ Pg) {
  // P: by-value copy in the kernarg segment (global pointers, scalar loads) for the inlined hot paths;
  // PG: the same block in HBM, handed by reference to the out-of-line cold routines.
  const KParams& PG = *Pg;
  const u32 part = P.part_map ? uni32(P.part_map[blockIdx.x]) : blockIdx.x;   // (a cycle may be split over two launches: KParams::part_map)
  const u32 tid = threadIdx.x, lane = tid & 63u;
  const u32 wave = uni32(tid >> 6);  // wave-uniform: the role split below is a scalar branch
  // (flat loads are sources of divergence for the compiler: make what steers the control flow uniform)
  const u32 qbeg = uni32(P.part_off[part]);
  const u32 nn = uni32(P.part_off[part + 1]) - qbeg;
  const u64 jbeg = uni64(P.pj_off[part]), jend = uni64(P.pj_off[part + 1]);
  if (jbeg >= jend) return;
  // a reservation's scheduler exists only while the reservation is active (JobScheduler.cpp:6643,6754-6759)
  const bool resv_part = part >= P.num_real_parts;
  if (resv_part) {
    const i64 rs = P.resv_se[2 * (part - P.num_real_parts)], re = P.resv_se[2 * (part - P.num_real_parts) + 1];
    if (!(rs <= P.now && P.now < re)) {
      for (u64 x = jbeg + tid; x < jend; x += kBlock) {
        const u32 orig = P.jobrec[x * kJobRecDwords + kJrOrig];
        P.o_start[orig] = 0;
        P.o_reason[orig] = 6;  // "Reservation Not Found"
      }
      return;
    }
  }
#ifdef CNS_HWID
  if (lane == 0 && part == 0) printf("wave %u hw_id %08x simd %u\n", wave, (unsigned)__builtin_amdgcn_s_getreg(63492), ((unsigned)__builtin_amdgcn_s_getreg(63492) >> 4) & 3u);
#endif

  __shared__ u64 s_wc[2][kRed];
  __shared__ u32 s_wp[2][kRed];
  __shared__ u64 s_tc[kRed];
  __shared__ u32 s_tp[kRed];
  __shared__ u64 s_pc[kRed];   // pre-scan of the NEXT job: per-wave A / T argmins over the nodes that
  __shared__ u32 s_pp[kRed];   // cannot change (everything but this job's round-0 winners)
  __shared__ u64 s_ptc[kRed];
  __shared__ u32 s_ptp[kRed];
  __shared__ u64 s_win_c[2];     // winners of the next job as merged by the worker: [0] = A, [1] = T
  __shared__ u32 s_win_p[2];
  __shared__ u32 s_on[4];        // scan summary of this job's T winner (fcpu, mw, gn), posted by its owner lane
  __shared__ u64 s_lc[(kWaves - 1) * kMultiK];  // multi-node jobs: per-wave sorted candidate lists (cost keys / slot codes)
  __shared__ u32 s_lp[(kWaves - 1) * kMultiK];
  __shared__ i64 s_nf[2 * kMultiK];             // ... and the next-fit exchange of the backfill fixed point
  __shared__ int s_mode;                        // ... worker -> scanners: 1 start-now list complete, 2 need res_total lists, 3 res_total list complete, 0 give up
  __shared__ int s_fl[8];   // [0] the worker's verdict; [4..7] the dip a rejected start-now candidate revealed (post_dip): slot code | seconds after now | cpus, mem | GRES counts
#define s_flag s_fl[0]
  __shared__ int s_r0;   // worker -> scanners: this job may be followed by the worker-side merge
  __shared__ int s_nupd;
  __shared__ UpdRec s_upd[kMaxUpd];
  __shared__ HeapEnt s_heap[kLdsHeap];
  __shared__ JobCtx s_job;
  __shared__ int s_ty_cpu[CNS_MAX_NODE_TYPES_DEV];  // per node type: what "completely free" looks like
  __shared__ u32 s_ty_m16[CNS_MAX_NODE_TYPES_DEV];
  __shared__ u32 s_ty_gn[CNS_MAX_NODE_TYPES_DEV];
  __shared__ GresDev s_gres;  // LDS copy of the GRES layout for the out-of-line helpers (their parameter block is read with FLAT loads: from LDS the table lookups of the exact test are ~10x closer than from HBM)
  // Worker-private cache of GetFeasibleResourceInNode(request, res_total) (:6354-6356): the allocation of a backfilled
  // job depends only on (cpu, GRES request, node type); computing it cost 4.1 k cycles of the serial chain per job.
  __shared__ AllocCacheEnt s_ac[kAllocCache];
  __shared__ u32 s_nme[kMaxNames], s_nmo[kMaxNames];  // GRES name masks in the split-nibble domain (even / odd nibbles as bytes)

  const Res ttot = lane < P.num_types ? P.type_total[lane] : res_zero();  // lane t holds node type t
  TypeLane tyl;
  tyl.cpu = ttot.cpu; tyl.mem = ttot.mem; tyl.ncores = cores_count(ttot);
  tyl.cnt = class_counts(ttot.gres, P.gres);
  int par = 0;

  if (wave == 0) {
    // =============================================================================================
    // WORKER
    // =============================================================================================
    if (lane < (u32)kRed && (lane == 0 || lane >= (u32)kWaves)) {  // exchange slots no scanner writes: identity
      s_wc[0][lane] = ~0ull; s_wc[1][lane] = ~0ull; s_wp[0][lane] = kNone; s_wp[1][lane] = kNone;
      s_tc[lane] = ~0ull; s_tp[lane] = kNone;
      s_pc[lane] = ~0ull; s_pp[lane] = kNone; s_ptc[lane] = ~0ull; s_ptp[lane] = kNone;
    }
    s_ty_cpu[lane] = clamp_cpu(ttot.cpu);
    s_ty_m16[lane] = mem_gib16(mem_mib_ceil(ttot.mem));
    s_ty_gn[lane] = nibbles_of(tyl.cnt);
    if (lane == 0) s_gres = P.gres;
    for (u32 x = lane; x < (u32)kAllocCache; x += 64) s_ac[x].type = kNone;
    if (lane < (u32)kMaxNames) {
      const u32 nb = nibbles_of(P.gres.name_bytes[lane] & 0x0F0F0F0F0F0F0F0Full);  // nibble g = 0xF if class g is in the name
      s_nme[lane] = nb & 0x0F0F0F0Fu;
      s_nmo[lane] = (nb >> 4) & 0x0F0F0F0Fu;
    }
    wg_barrier();  // type tables visible to the scanners
    WorkerShared sh;
    sh.wc = s_wc; sh.wp = s_wp; sh.flag = &s_flag; sh.nupd = &s_nupd; sh.upd = s_upd; sh.heap = s_heap; sh.part = part;
    HeapEnt* const gheap = P.heap + qbeg + part;
    // The worker is the serial chain of the whole partition and shares its SIMD with three scanner waves
    // that pre-scan the next job at the same time: let its instructions issue first.
#ifndef CNS_NO_PRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    u32 raw = fetch_job(P, jbeg);       // record of the job being processed
    u32 raw_n = jbeg + 1 < jend ? fetch_job(P, jbeg + 1) : 0u;  // next job's record: in flight during this job
    bool pre_valid = false;   // winners of this job already known from the previous iteration's merge
    u64 wc = ~0ull, tc = ~0ull;
    u32 wcode = kNone, tcode = kNone;
    u64 ji = jbeg;
    while (ji < jend) {
      PROF_T(p0);
      const FastJob F = make_fast_job(P, raw);
      const bool simple = !(F.flags & kJfExclusive) && F.ntasks == F.k;  // ntasks == node_num on shared nodes
      // partitions that share nodes / a cycle with preemption: everything through the general path — but for the jobs of such a
      // cycle that can preempt nobody (inline_in_preempt_cycle)
      const bool shared_nodes = general_path_job(P.general_only != 0, P.sib_off != nullptr, F.flags, F.k, F.ntasks != F.k, F.tmin == 1);
      const bool fast = simple && F.k == 1 && F.tmin == 1 && !shared_nodes;
      // A job of the inline path's shape that MAY preempt (a cycle with preemption): TryPreempt_ only comes after "start now" has
      // failed (JobScheduler.cpp:6140-6143), so its phase A runs inline as well — the owner update goes where the scanners of a
      // general-path job look (the HBM list) — and only what is left (TryPreempt_, Backfill_) goes out of line, from the round reached.
      const bool fast_a = P.general_only && !P.sib_off && simple && F.k == 1 && F.tmin == 1 && (F.flags & kJfMayPreempt) != 0;
      const u64 wsmode = P.general_only ? ~0ull : 0ull;   // (the scanners' keys are in signed form in such a cycle: cost_key_m)

      if (!pre_valid) {
        wg_barrier();  // B1 (round 0): A and T argmins published by the scanners
        wc = s_wc[par][lane & (kRed - 1)];
        wcode = s_wp[par][lane & (kRed - 1)];
        tc = s_tc[lane & (kRed - 1)];
        tcode = s_tp[lane & (kRed - 1)];
        reduce16(wc, wcode);
        reduce16(tc, tcode);
        par ^= 1;
        wc = uni64(wc); wcode = uni32(wcode); tc = uni64(tc); tcode = uni32(tcode);
      }
      PROF_T(p1);
      PROF_ADD(0, p0, p1);  // worker: wait for the scanners (only when the pre-scan could not be used)
      bool round0 = true;       // this job may be followed by the worker-side merge
      NodeSum cn, on;           // scan summaries after this job: the committed / examined node, the other winner
      cn.code = kNone; cn.cost = 0; cn.len = 0; cn.type = 0; cn.fcpu = 0; cn.fmem = 0; cn.fcnt = 0;
      on = cn;

      bool divert = !fast && !fast_a;   // leave the inline path (multi-node / general / exclusive / long time map)
      bool have_on = false;     // `on` is the T winner: its summary sits in s_on after B2
      if (fast || fast_a) {
        // The T winner's summary is needed for the merge when this job commits on the A winner; its owner
        // lane posts it in LDS (s_on) before B2 — no HBM round trip on the serial chain.
        have_on = wcode != kNone && tcode != kNone && tcode != wcode;
        if (have_on) { on.code = tcode; on.cost = tc; }
        // ---- fast path, Phase A: start now (GetNodesAndTrySchedule_, JobScheduler.cpp:6188-6333) ------
        bool done = false;
        while (wcode != kNone) {
          PROF_T(a0);
          const u32 q = qbeg + slot_of_code(wcode);
          int code = 0;
          // one coalesced read: header (broadcast) + entry `lane` of the time map
          NodeHdr* hd; NodeHdr h; TlEntry e;
          load_block(P, q, lane, hd, h, e);
          if (h.len > 64) { divert = true; break; }  // long time map: the general routines take over from here
          PROF_T(a1);
          PROF_ADD(1, a0, a1);  // node block load
          Res f, m;
          bool ok = false;
          // :6274 wants GetFeasibleResourceInNode(res_avail) to succeed, :6285 the same on the window minimum m.
          // m is res_avail folded with Ckmin (:6278-6283): cpu and mem are minima, GRES slots and (when non-empty)
          // core ids are subsets.  So success on m implies the cpu, mem and GRES tests of :6274; the one test of
          // :6274 that m does not imply is the core-id count of res_avail (:534) when m's core set came out empty.
          PROF_T(a1w);
          m = uni_res(window_min_regs(e, lane < h.len, h.avail0, F.E));   // :6278-6283
          PROF_T(a1x);
          PROF_ADD(22, a1, a1w);   // phase A: (nothing left before the window-min)
          PROF_ADD(23, a1w, a1x);  // phase A: window-min
          ok = feasible(F.mv, m, f, P.gres);                             // get_max_tasks(min_res) > 0, :6285
          if (ok) {
            const i64 req_int = F.mv.cpu / 256;
            const u32 nc0 = cores_count(h.avail0);
            if (req_int * 256 == F.mv.cpu && nc0 != 0 && nc0 < (u32)req_int) ok = false;   // :528-534 on res_avail
          }
          PROF_T(a2);
          PROF_ADD(2, a1, a2);  // window-min + feasibility
          if (ok) {  // tpn_min == 1: the minimum view is the 1-task view, f is the allocation (:6312-6320)
            commit_single_regs(P, F.L, F.orig, F.poff, hd, h, e, q, wcode, cost_of_key_m(wc, wsmode), f,
                               P.now, 0, lane, fast_a ? P.g_upd + qbeg : (UpdRec*)s_upd, &s_nupd, cn, PG, qbeg);
            if (fast_a) __threadfence_block();   // (the record went through HBM)
            cn.cost = cost_key_m(__longlong_as_double((long long)cn.cost), wsmode);   // (the merge below compares it with the scanners' keys)
            code = 2;
          } else if (NPL <= kSelDipMaxNpl) {
            const bool dipped = post_dip(P, P.gres, s_fl, wcode, e, h.len, F.mv, F.E, lane, m);   // what this window tripped over, for the node's owner lane
            if (dipped) { PROF_DIP(28); }
#ifdef CNS_PROF_DIP
            {   // why was it rejected?  front entry fails the exact test: 29 by cpu, 30 by memory, 31 otherwise; 27: front fine, no single entry fails by counts
              Res ftmp;
              const bool front_ok = feasible(F.mv, h.avail0, ftmp, P.gres);
              if (!front_ok) {
                if (F.mv.cpu > h.avail0.cpu) { PROF_DIP(29); }
                else if (F.mv.mem > h.avail0.mem) { PROF_DIP(30); }
                else { PROF_DIP(31); }
              }
              if (front_ok && !dipped) { PROF_DIP(27); }
            }
#endif
          }
          PROF_T(a3);
          PROF_ADD(3, a2, a3);  // commit
          if (lane == 0) { s_flag = code; s_r0 = round0 ? 1 : 0; }
          wg_barrier();  // B2: verdict (and, on success, the owner updates) visible to the scanners
          if (code == 2) { done = true; break; }
          PROF_CNT(14);         // rejected candidate
          round0 = false;
          wg_barrier();  // B1 of the next round
          wc = s_wc[par][lane & (kRed - 1)];
          wcode = s_wp[par][lane & (kRed - 1)];
          reduce16(wc, wcode);
          par ^= 1;
          wc = uni64(wc); wcode = uni32(wcode);
        }
        if (done) {
          PROF_CNT(11);
        } else if (fast_a) {
          divert = true;   // nothing starts it now: TryPreempt_ / Backfill_ out of line (worker_job_slow enters with no start-now candidate left)
        } else if (!divert) {
          // ---- fast path, Phase B: the first node in cost order whose res_total fits (the T argmin of
          // round 0), earliest start on its time map (JobScheduler.cpp:6335-6368, Backfill_ :6371-6376) ----
          PROF_T(b0);
          int code = 0;
          on.code = kNone;
          have_on = false;
          NodeHdr* hd = nullptr; NodeHdr h; TlEntry e;
          i64 first_resv = kInf;
          if (tcode != kNone) {
            if (!resv_part) first_resv = P.first_resv[qbeg + slot_of_code(tcode)];  // in flight with the block
            load_block(P, qbeg + slot_of_code(tcode), lane, hd, h, e);
            if (h.len > 64) divert = true;
          }
          PROF_T(b0l);
          PROF_ADD(8, b0, b0l);  // phase B: node block load
          if (!divert) {
            if (tcode != kNone) {
              const u32 q = qbeg + slot_of_code(tcode);
              Res alloc = res_zero();
              {
                u32 hs = (u32)F.mv.cpu * 0x9E3779B1u ^ (u32)(F.mv.gspec * 0x85EBCA6B5BD1E995ull >> 29) ^ F.mv.gtot * 0xC2B2AE35u ^ h.type * 0x27D4EB2Fu;
                hs = (hs ^ (hs >> 15)) & (kAllocCache - 1);
                const AllocCacheEnt c = s_ac[hs];
                const bool hit = uni32((c.type == h.type && c.cpu == F.mv.cpu && c.gspec == F.mv.gspec && c.gtot == F.mv.gtot) ? 1u : 0u) != 0;
                if (hit) {
                  alloc.cpu = F.mv.cpu; alloc.mem = F.mv.mem;
                  alloc.clo = uni64(c.clo); alloc.chi = uni64(c.chi); alloc.gres = uni64(c.gres);
                  alloc.c2 = uni64(c.c2); alloc.c3 = uni64(c.c3);
                } else {
                  if (!feasible(F.mv, h.total, alloc, P.gres)) {  // :6354-6356; cannot fail: the T argmin only ranks nodes whose res_total fits
                    if (lane == 0) set_fault(P, 3, F.orig, h.node, 0);
                  } else if (lane == 0) {
                    AllocCacheEnt w;
                    w.cpu = F.mv.cpu; w.gspec = F.mv.gspec; w.gtot = F.mv.gtot; w.type = h.type;
                    w.clo = alloc.clo; w.chi = alloc.chi; w.gres = alloc.gres; w.c2 = alloc.c2; w.c3 = alloc.c3;
                    s_ac[hs] = w;
                  }
                }
              }
              PROF_T(b0f);
              PROF_ADD(16, b0l, b0f);  // phase B: allocation against res_total
              const i64 st = next_fit_regs(e, h.len, alloc, F.L, P.now, lane);
              PROF_T(b0n);
              PROF_ADD(9, b0l, b0n);  // phase B: allocation against res_total + next fit
              // the node as the scanners see it if nothing is committed
              const Res e0 = rl_res(e.r, 0);
              cn.code = tcode; cn.cost = tc; cn.len = h.len; cn.type = h.type;
              cn.fcpu = clamp_cpu(e0.cpu); cn.fmem = mem_mib_ceil(e0.mem); cn.fcnt = class_counts(e0.gres, P.gres);
              if (st != kInf && st - P.now <= P.max_window) {          // kAlgoMaxTimeWindow, JobScheduler.h:815
                int reason = 0;
                if (st != P.now) {  // :6797-6831
                  if (!resv_part && first_resv < P.now + F.L) reason = 3;  // "Resource Reserved" (:6799-6806)
                  else reason = res_le(alloc, h.avail0) ? 1 /*Priority*/ : 2 /*Resource*/;
                }
                PROF_T(b0c);
                commit_single_regs(P, F.L, F.orig, F.poff, hd, h, e, q, tcode, cost_of_key_m(tc, wsmode),
                                   alloc, st, reason, lane, s_upd, &s_nupd, cn, PG, qbeg);
                cn.cost = cost_key_m(__longlong_as_double((long long)cn.cost), wsmode);
                PROF_T(b0d);
                PROF_ADD(10, b0c, b0d);  // phase B: commit
                code = 2;
              }
            }
            if (code == 0 && lane == 0) { P.o_start[F.orig] = 0; P.o_reason[F.orig] = 2; }  // "Resource", :6768
            if (lane == 0) { s_flag = code; s_r0 = round0 ? 1 : 0; }
            PROF_T(b1);
            PROF_ADD(4, b0, b1);  // backfill + commit
            PROF_CNT(12);
            wg_barrier();  // B3
          }
        }
      }
      if (divert) {
        // out-of-line continuation from the current round: same barrier schedule, general code
        round0 = false;
        if (lane == 0) s_r0 = 0;
        PROF_T(d0);
        job_to_lds(PG, ji, raw, &s_job);
        PROF_T(d1);
        PROF_ADD(24, d0, d1);  // job record -> LDS
        if (simple && F.k > 1 && F.k <= (u32)kMultiK && F.tmin == 1 && !shared_nodes) {
          // ---- multi-node job, parallel protocol: every scanner wave lists its k best candidates, the
          // worker merges the 15 sorted lists into the first k nodes in cost order, then the candidates are
          // verified and committed in parallel, one helper wave per node ----------------------------------
          // k-way merge: lane w < 15 holds the head of wave w's list
          auto merge_lists = [&]() -> u32 {
            u32 idx = 0, n = 0;
            u64 hc = lane < kWaves - 1 ? s_lc[lane * kMultiK] : ~0ull;
            u32 hp = lane < kWaves - 1 ? s_lp[lane * kMultiK] : kNone;
            for (u32 i = 0; i < F.k; ++i) {
              u64 c = hc;
              u32 pc = hp;
              reduce16(c, pc);
              if (pc == kNone) break;
              if (lane == 0) {
                HeapEnt x; x.ntasks = 0; x.p = pc; x.node = 0; x.pad = 0;
                x.cost = cost_of_key_m(c, wsmode); x.res = res_zero();
                s_heap[i] = x;
              }
              ++n;
              if (lane == owner_wave(pc) - 1u) {  // the winning wave's list advances
                ++idx;
                hc = idx < F.k ? s_lc[lane * kMultiK + idx] : ~0ull;
                hp = idx < F.k ? s_lp[lane * kMultiK + idx] : kNone;
              }
            }
            return n;
          };
          bool fallback = false;
          wg_barrier();  // M1: start-now lists posted
          u32 n = merge_lists();
          if (lane == 0) s_mode = n == F.k ? 1 : 2;
          PROF_T(d2);
          PROF_ADD(25, d1, d2);  // start-now lists + merge
          wg_barrier();  // M2
          if (n == F.k) {  // k start-now candidates exist (:6294-6297 if their exact tests pass)
            if (multi_verify_commit(PG, &s_gres, &s_job, s_heap, kWaves - 1, (u32)(kWaves - 1) < F.k, qbeg, s_upd, &s_nupd)) {
              if (lane == 0) { P.o_start[F.orig] = P.now; P.o_reason[F.orig] = 0; }  // :6326
              if (P.pre.enabled) pre_join_multi(PG, s_job, s_heap, qbeg, P.now + F.L);
              PROF_CNT(30);
            } else {
              fallback = true;  // a candidate failed its exact test (rare): redo this job sequentially
            }
            PROF_T(d3);
            PROF_ADD(26, d2, d3);  // helpers verify + commit
          } else {  // fewer than k nodes can start it now -> first k by res_total + backfill (:6335-6376)
            wg_barrier();  // M5: res_total lists posted
            n = merge_lists();
            if (lane == 0) s_mode = n == F.k ? 3 : 0;
            PROF_T(d5);
            PROF_ADD(28, d2, d5);  // res_total lists + merge
            wg_barrier();  // M6
            if (n == F.k) {
              int reason = 0;
              const i64 st = multi_backfill_par(PG, &s_gres, &s_job, s_heap, kWaves - 1, (u32)(kWaves - 1) < F.k, qbeg, s_upd, &s_nupd, s_nf, &reason);
              PROF_T(d6);
              PROF_ADD(29, d5, d6);  // common earliest start + commit
              PROF_CNT(31);
              if (lane == 0) {
                if (st != kInf) { P.o_start[F.orig] = st; P.o_reason[F.orig] = (uint8_t)reason; }
                else { P.o_start[F.orig] = 0; P.o_reason[F.orig] = 2; }  // "Resource", :6768
              }
              if (P.pre.enabled && st != kInf) pre_join_multi(PG, s_job, s_heap, qbeg, st + F.L);
            } else if (lane == 0) {
              P.o_start[F.orig] = 0; P.o_reason[F.orig] = 2;  // not even k nodes fit res_total (:6335-6343)
            }
          }
          if (fallback) par = P.sib_off ? worker_job_slow(PG, sh, &s_job, par, wc, wcode, tc, tcode, qbeg, gheap)   // (sibling slots among the updates)
                                        : worker_job_multi(PG, sh, &s_job, par, wc, wcode, tc, tcode, qbeg);
          PROF_T(p8);
          PROF_ADD(6, d0, p8);
          PROF_CNT(15);
        } else if (simple && F.k > 1 && F.k <= (u32)kMaxUpd && F.tmin == 1 && !shared_nodes) {
          par = worker_job_multi(PG, sh, &s_job, par, wc, wcode, tc, tcode, qbeg);  // 15 < k <= 32: sequential protocol
          PROF_T(p8);
          PROF_ADD(6, d0, p8);
          PROF_CNT(15);
        } else {
          par = worker_job_slow(PG, sh, &s_job, par, wc, wcode, tc, tcode, qbeg, gheap);
          PROF_T(p9);
          PROF_ADD(5, d0, p9);
          PROF_CNT(13);
        }
      }

      // ---- next job: if the scanners' pre-scan is usable, merge the (at most two) nodes this job could
      // have changed into it right here — the scanners are not on the critical path ------------------------
      PROF_T(m0);
      if (ji + 1 >= jend) break;
      raw = raw_n;
      raw_n = ji + 2 < jend ? fetch_job(P, ji + 2) : 0u;
      const u32 nflags = rl32(raw, kJrFlags);
      const bool nv = fast && round0 && !(nflags & (kJfExclusive | kJfIncl | kJfExcl)) && !P.sib_off;   // (no pre-scan on shared nodes: a commit changes more rows than the two winners'; in a cycle with preemption the jobs of the inline path have it — cn / on carry keys in the scanners' signed form)
      if (nv) {
        const FastJob Fn = make_fast_job(P, raw);
        u64 ac = s_pc[lane & (kRed - 1)];
        u32 ap = s_pp[lane & (kRed - 1)];
        u64 tcs = s_ptc[lane & (kRed - 1)];
        u32 tp = s_ptp[lane & (kRed - 1)];
        reduce16(ac, ap);
        reduce16(tcs, tp);
        ac = uni64(ac); ap = uni32(ap); tcs = uni64(tcs); tp = uni32(tp);
        const u64 tyok = jr64(raw, kJdTyok);
        if (have_on) {  // expand the tile words of the T winner (saturated fields stay conservative)
          const u32 ocpu = uni32(s_on[0]), omw = uni32(s_on[1]), ogn = uni32(s_on[2]);
          on.fcpu = (int)ocpu;
          on.len = (omw >> 16) & 0x3FFu;
          on.type = omw >> 26;
          on.fmem = (omw & 0xFFFFu) == 0xFFFFu ? 0xFFFFFFFFu : ((omw & 0xFFFFu) << 10);
          on.fcnt = bytes_of_nibbles(ogn);
        }
        bool b, a;
        eval_node(P, Fn.mv, Fn.flags, tyok, cn, b, a);
        if (a && (cn.cost < ac || (cn.cost == ac && cn.code < ap))) { ac = cn.cost; ap = cn.code; }
        if (b && (cn.cost < tcs || (cn.cost == tcs && cn.code < tp))) { tcs = cn.cost; tp = cn.code; }
        eval_node(P, Fn.mv, Fn.flags, tyok, on, b, a);
        if (a && (on.cost < ac || (on.cost == ac && on.code < ap))) { ac = on.cost; ap = on.code; }
        if (b && (on.cost < tcs || (on.cost == tcs && on.code < tp))) { tcs = on.cost; tp = on.code; }
        wc = ac; wcode = ap; tc = tcs; tcode = tp;
        if (lane == 0) { s_win_c[0] = wc; s_win_p[0] = wcode; s_win_c[1] = tc; s_win_p[1] = tcode; }
        wg_barrier();  // B1': winners of the next job visible to the scanners
      }
      PROF_T(m1);
      PROF_ADD(7, m0, m1);  // worker: next-job decode + merge
      pre_valid = nv;
      ++ji;
    }
  } else {
    // =============================================================================================
    // SCANNERS — register-resident node tile: slot p = r*960 + t, code = r<<10 | t
    // =============================================================================================
    const bool idle = wave == (u32)kIdleWave;  // the worker's SIMD mate: no nodes, follows the barriers, helps with multi-node jobs
    const u32 t = idle ? 1023u : (wave - 1u - (wave > (u32)kIdleWave ? 1u : 0u)) * 64u + lane;
    using RM = typename RowMask<(NPL > 32)>::type;  // one bit per row of the lane
    const RM kOne = 1;
    double cost[NPL];
    int fcpu[NPL];
    u32 mw[NPL];   // fmem GiB (16) | len (10) << 16 | type (6) << 26 ; len = 1023 marks "no node"
    u32 gn[NPL];   // per-class free-slot count, 4 bits each
    // the row's dip (post_dip; KParams::dip_* as k_init_nodes found it): seconds after now (~0: none) | cpus, mem | GRES counts
    constexpr bool kDip = NPL <= kSelDipMaxNpl;
    constexpr int kDipRows = kDip ? NPL : 1;
    u32 dpa[kDipRows], dpg[kDipRows];   // pack_dip(seconds, cpus | mem) and the GRES counts: two registers per row (three cost k_select<19> 16 % on C4)
#pragma unroll
    for (int r = 0; r < NPL; ++r) {
      const u32 p = (u32)r * kScan + t;
      if (!idle && p < nn) {
        const u32 q = qbeg + p;
        const NodeHdr* hd = hdr_of(P, q);
        cost[r] = P.cost[q];
        fcpu[r] = P.f_cpu[q];
        mw[r] = mem_gib16(P.f_mem[q]) | (hd->len << 16) | ((u32)P.slot_type[q] << 26);
        gn[r] = nibbles_of(P.f_cnt[q]);
        if (kDip) { dpa[r] = pack_dip(P.dip_t[q], P.dip_cm[q]); dpg[r] = P.dip_g[q]; }
      } else {
        cost[r] = 0.0; fcpu[r] = 0; mw[r] = 1023u << 16; gn[r] = 0;
        if (kDip) { dpa[r] = kDipNone; dpg[r] = 0; }
      }
    }
    wg_barrier();  // type / name tables written by the worker

    const u32 maxlen = P.max_jobs_per_node;
    const u32 G8 = 0x80808080u;
    // Static part of the "may host the job at all" filter, one bit per row of the lane:
    //   okbits: the row holds a node and its time map has room (len < kAlgoMaxJobNumPerNode, :6194);
    //           refreshed by the owner update, the only place a length changes
    //   wave_types: node types present in this wave's tile (a job that fits all of them needs no per-row
    //           type test)
    RM okbits = 0;
    u64 wave_types = 0;
#pragma unroll
    for (int r = 0; r < NPL; ++r) {
      const u32 len = (mw[r] >> 16) & 0x3FFu;
      okbits |= (RM)(len < maxlen ? 1u : 0u) << r;  // "no node" rows carry len = 1023
      if (len != 1023u) wave_types |= 1ull << (mw[r] >> 26);
    }
    wave_types = wave_or_u64(wave_types);
    // Partitions that share nodes run as ONE group: a row then belongs to one member partition (byte r of tg) and a job
    // only sees the rows of its own partition (its tag sits in bits 8..15 of the job flags).
    u32 tg[(NPL + 3) / 4];
#pragma unroll
    for (int x = 0; x < (NPL + 3) / 4; ++x) tg[x] = 0;
    if (P.slot_tag) {
#pragma unroll
      for (int r = 0; r < NPL; ++r) {
        const u32 p = (u32)r * kScan + t;
        if (!idle && p < nn) tg[r / 4] |= (u32)P.slot_tag[qbeg + p] << (8 * (r % 4));
      }
    }

    // What a scanner keeps of a job: the request side of the filters, ~10 scalars (the full JobCtx only
    // exists in LDS for the out-of-line paths).  Decoded from the lane-striped record with v_readlane.
    struct ScanJob {
      u32 flags, k;
      u32 shape;   // bit 0: ntasks != node_num (general), bit 1: tpn_min == 1, bit 2: request is satisfiable at all
      int rc32;    // min-view cpu, clamped to the 32-bit front summary
      u32 rm16;    // min-view mem in GiB, rounded down, saturating
      u32 rq;      // specified GRES counts per class, nibbles saturating at 15 (like the node side)
      u32 gtot;    // untyped totals per name (bytes)
      u32 gmode;   // 0 no GRES; bit 0: one specified class, bit 1: one untyped total (1..3 = short tests); 4 general
      u32 gsel;    // nibble shift of that class | index of that name << 8
      u32 gneed;   // its count | the total << 8, both saturated at 15
      u32 loff;    // time limit in seconds, saturating: an entry t seconds after now lies in a start-now window iff t < loff (:6279)
    };
    auto decode = [&](u32 raw, u64& tyok) {
      ScanJob S;
      S.flags = rl32(raw, kJrFlags);
      S.k = rl32(raw, kJrK);
      S.shape = rl32(raw, kJdShape);
      S.rc32 = (int)rl32(raw, kJdRc32);
      S.rm16 = rl32(raw, kJdRm16);
      S.rq = rl32(raw, kJdRq);
      S.gtot = rl32(raw, kJrGtot);
      S.gmode = rl32(raw, kJdGmode);
      S.gsel = rl32(raw, kJdGsel);
      S.gneed = rl32(raw, kJdGneed);
      {
        const u64 L = jr64(raw, kJrL);
        S.loff = L >= 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)L;
      }
      tyok = jr64(raw, kJdTyok);
      return S;
    };

    // Filters of job S on ONE node given its tile words:
    //   b: the node may host the job at all (len < 1000 :6194, res_total fits :6222)
    //   a: ... and may start it now (front filter: necessary for :6274-6285 — the entry at `now` is in every window)
    // gme / gmo: the name masks of mode 2 (even / odd nibbles), hoisted by the caller.
    auto row_pred = [&](const ScanJob& S, u64 tyok, u32 gme, u32 gmo, u32 w, int fc, u32 g, bool& b, bool& a) {
      b = ((S.shape & 4u) != 0) & (((tyok >> (w >> 26)) & 1ull) != 0) & (((w >> 16) & 0x3FFu) < maxlen);
      a = b & (S.rc32 <= fc) & (S.rm16 <= (w & 0xFFFFu));
      if (S.gmode & 1u) a = a & (((g >> (S.gsel & 0xFFu)) & 15u) >= (S.gneed & 0xFFu));
      if (S.gmode & 2u) {
        const u32 ce = g & 0x0F0F0F0Fu, co = (g >> 4) & 0x0F0F0F0Fu;
        a = a & (__builtin_amdgcn_sad_u8(ce & gme, 0u, __builtin_amdgcn_sad_u8(co & gmo, 0u, 0u)) >= (S.gneed >> 8));
      }
      if (S.gmode == 4) {
        u32 gr = g;
        asm volatile("" : "+v"(gr));  // pins the GRES arithmetic inside this (job-uniform) branch
        const u32 ce = gr & 0x0F0F0F0Fu, co = (gr >> 4) & 0x0F0F0F0Fu;
        const u32 rqe = S.rq & 0x0F0F0F0Fu, rqo = (S.rq >> 4) & 0x0F0F0F0Fu;
        a = a & ((((ce | G8) - rqe) & G8) == G8) & ((((co | G8) - rqo) & G8) == G8);
#pragma unroll
        for (int n = 0; n < kMaxNames; ++n) {
          const u32 tot = (S.gtot >> (8 * n)) & 0xFFu;
          const u32 have = __builtin_amdgcn_sad_u8(ce & s_nme[n], 0u, __builtin_amdgcn_sad_u8(co & s_nmo[n], 0u, 0u));
          a = a & ((tot == 0) | (have >= (tot > 15u ? 15u : tot)));
        }
      }
    };

    // argmin of (cost, code) over the lane's nodes whose bit is set in `mask`; ties keep the lower r
    const u64 smode = P.general_only ? ~0ull : 0ull;   // signed cost keys in a cycle with preemption (cost_key_m)
    auto lane_argmin = [&](RM mask, u64& bc, u32& bp) {
      bc = ~0ull;
      u32 br = 0xFFu;
#pragma unroll
      for (int r = 0; r < NPL; ++r) {
        const u64 ck = cost_key_m(cost[r], smode);
        const bool take = ((mask >> r) & 1u) & (ck < bc);
        bc = take ? ck : bc;
        br = take ? (u32)r : br;
      }
      bp = br == 0xFFu ? kNone : ((br << 10) | t);
    };

    // Candidate sets of one job as per-lane bitmasks + their lane-local argmins.  Nothing the filters look
    // at changes during a job (commits happen at its end), so this runs ONCE per job.
    // `skip` masks out nodes whose state is about to change (speculative pre-scan, see below).
    // (always_inline: with six row loops inside, the inliner otherwise leaves the second call out of line and the tile goes to scratch)
    auto scan_job = [&](const ScanJob& S, u64 sji, u64 tyok, RM skip, RM& bmask, RM& amask, u64& ac, u32& ap,
                        u64& tcs, u32& tp) __attribute__((always_inline)) {
      if (idle) { bmask = 0; amask = 0; ac = ~0ull; ap = kNone; tcs = ~0ull; tp = kNone; return; }
      // bmask in one go: static bits & type bits & not skipped
      RM tybits = (RM)(((u64)1 << NPL) - 1ull);
      if ((tyok & wave_types) != wave_types) {  // some type present here cannot host the job (uniform, rare)
        tybits = 0;
#pragma unroll
        for (int r = 0; r < NPL; ++r) tybits |= (RM)((tyok >> (mw[r] >> 26)) & 1ull) << r;
      }
      if (P.slot_tag) {   // (uniform) only the rows of the job's own partition
        const u32 jt = (S.flags >> 8) & 0xFFu;
        RM tb = 0;
#pragma unroll
        for (int r = 0; r < NPL; ++r) tb |= (RM)(((tg[r / 4] >> (8 * (r % 4))) & 0xFFu) == jt ? 1u : 0u) << r;
        tybits &= tb;
      }
      const RM bl = (S.shape & 4u) ? (RM)(okbits & tybits & ~skip) : (RM)0;
      u32 gme = 0, gmo = 0;
      if (S.gmode & 2u) { gme = uni32(s_nme[S.gsel >> 8]); gmo = uni32(s_nmo[S.gsel >> 8]); }
      RM am = 0;
      ac = ~0ull; tcs = ~0ull;
      u32 ar = 0xFFu, tr = 0xFFu;
      // the row loop, specialised on the shape of the GRES request (job-uniform): 0 none, 1 short tests, 4 general
      auto rows = [&](auto mode, auto sparse) __attribute__((always_inline)) {
        constexpr u32 M = decltype(mode)::value;
        // a group that shares nodes: a job sees the rows of its own partition only and the slots are laid out partition by partition —
        // whole rows of the wave are not its own (C4all: 37 rows, 18 or 19 of them the job's; 1 966 -> 1 719 ms)
        constexpr bool kSparse = decltype(sparse)::value != 0;
        const u32 sh1 = S.gsel & 0xFFu, need1 = S.gneed & 0xFFu, need2 = S.gneed >> 8;
        const u32 rqe = S.rq & 0x0F0F0F0Fu, rqo = (S.rq >> 4) & 0x0F0F0F0Fu;
        // the request in pack_dip's units: whole cpus and GiB rounded down, capped like the dip's; the window in 16 s units, rounded down
        const u32 rcpus = (u32)S.rc32 >> 8;
        const u32 rc8 = rcpus > 255u ? 255u : rcpus, rm8 = S.rm16 > 255u ? 255u : S.rm16, lq = (S.loff >> 4) > 0xFFFFu ? 0xFFFFu : (S.loff >> 4);
        // the GRES side of the request against one set of per-class counts (the row's front, the row's dip)
        auto gfit = [&](u32 g) -> bool {
          bool ok = true;
          if (M == 1) {
            if (S.gmode & 1u) ok = ok & (((g >> sh1) & 15u) >= need1);
            if (S.gmode & 2u) {
              const u32 ce = g & 0x0F0F0F0Fu, co = (g >> 4) & 0x0F0F0F0Fu;
              ok = ok & (__builtin_amdgcn_sad_u8(ce & gme, 0u, __builtin_amdgcn_sad_u8(co & gmo, 0u, 0u)) >= need2);
            }
          }
          if (M == 4) {
            const u32 ce = g & 0x0F0F0F0Fu, co = (g >> 4) & 0x0F0F0F0Fu;
            ok = ok & ((((ce | G8) - rqe) & G8) == G8) & ((((co | G8) - rqo) & G8) == G8);
#pragma unroll
            for (int n = 0; n < kMaxNames; ++n) {
              const u32 tot = (S.gtot >> (8 * n)) & 0xFFu;
              const u32 have = __builtin_amdgcn_sad_u8(ce & s_nme[n], 0u, __builtin_amdgcn_sad_u8(co & s_nmo[n], 0u, 0u));
              ok = ok & ((tot == 0) | (have >= (tot > 15u ? 15u : tot)));
            }
          }
          return ok;
        };
#pragma unroll
        for (int r = 0; r < NPL; ++r) {
          const bool b = ((bl >> r) & 1u) != 0;
          if (kSparse && __ballot(b) == 0ull) continue;   // (uniform) nobody's row r may host the job
          bool a = b & (S.rc32 <= fcpu[r]) & (S.rm16 <= (mw[r] & 0xFFFFu));  // the entry at `now` is in every window
          if (M != 0) a = a & gfit(gn[r]);
          if (kDip) {   // a window that reaches the row's dip must fit the dip too: a second necessary condition
            bool fits = (rc8 <= ((dpa[r] >> 8) & 0xFFu)) & (rm8 <= (dpa[r] & 0xFFu));
            if (M != 0) fits = fits & gfit(dpg[r]);
            a = a & (((dpa[r] >> 16) >= lq) | fits);
          }
          const u64 ck = cost_key_m(cost[r], smode);
          const bool ta = a & (ck < ac);
          ac = ta ? ck : ac;
          ar = ta ? (u32)r : ar;
          const bool tt = b & (ck < tcs);
          tcs = tt ? ck : tcs;
          tr = tt ? (u32)r : tr;
          am |= (RM)(a ? 1u : 0u) << r;
        }
      };
      if (P.slot_tag) {
        if (S.gmode == 0) rows(ModeTag<0>{}, ModeTag<1>{});
        else if (S.gmode < 4) rows(ModeTag<1>{}, ModeTag<1>{});
        else rows(ModeTag<4>{}, ModeTag<1>{});
      } else {
        if (S.gmode == 0) rows(ModeTag<0>{}, ModeTag<0>{});
        else if (S.gmode < 4) rows(ModeTag<1>{}, ModeTag<0>{});
        else rows(ModeTag<4>{}, ModeTag<0>{});
      }
      bmask = bl; amask = am;
      ap = ar == 0xFFu ? kNone : ((ar << 10) | t);
      tp = tr == 0xFFu ? kNone : ((tr << 10) | t);
      if (S.flags & kJfExclusive) {  // exclusive: the node must be completely free now (necessary for :6251-6257)
        amask = 0;
#pragma unroll
        for (int r = 0; r < NPL; ++r) {
          const u32 w = mw[r];
          const u32 ty = w >> 26;
          const bool a = ((bmask >> r) & 1u) & (fcpu[r] >= s_ty_cpu[ty]) & ((w & 0xFFFFu) >= s_ty_m16[ty]) &
                         (gn[r] == s_ty_gn[ty]);
          amask |= (RM)(a ? 1u : 0u) << r;
        }
        lane_argmin(amask, ac, ap);
      }
      if (S.flags & (kJfIncl | kJfExcl)) {  // included / excluded node lists (rare)
        const RM lm = (RM)list_mask(Pg, S.flags, sji, bmask, qbeg + t, (u32)NPL);
        bmask &= lm;
        amask &= lm;
        lane_argmin(amask, ac, ap);
        lane_argmin(bmask, tcs, tp);
      }
    };

    u32 raw = fetch_job(P, jbeg);
    u64 typeok;
    ScanJob J = decode(raw, typeok);
    if (jbeg + 1 < jend) raw = fetch_job(P, jbeg + 1);

    RM bmask = 0, amask = 0;
    u64 ac = ~0ull, tcs = ~0ull;
    u32 ap = kNone, tp = kNone;
    // state carried from the previous iteration's pre-scan
    RM bmask_n = 0, amask_n = 0;
    bool pre_valid = false;  // the winners of this job came from the worker's merge of the pre-scan
    u64 wc = ~0ull, tc = ~0ull;
    u32 wcode = kNone, tcode = kNone;

    // `up` (slot code) is wave-uniform: the row select is a scalar branch, only the owner lane writes
    auto apply_upd = [&](const UpdRec& u, u32 up) {
      const int rr = (int)(up >> 10);
      const bool own = (up & 1023u) == t;
      const double ucost = u.cost;
      const u32 ulen = u.len, ufront = u.has_front & 1u;
      const bool keep_cost = (u.has_front & 2u) != 0;   // the slot of another partition on a shared node
      const bool released = (u.has_front & kUpdReleased) != 0;   // TryPreempt_ gave resources back on this node: the row's dip is void
      const int ucpu = u.fcpu;
      const u32 um16 = mem_gib16(u.fmem), ugn = nibbles_of(u.fcnt);
#pragma unroll
      for (int r = 0; r < NPL; ++r)
        if (r == rr) {
          u32 w = (mw[r] & ~(0x3FFu << 16)) | (ulen << 16);
          if (ufront) w = (w & ~0xFFFFu) | um16;
          cost[r] = (own && !keep_cost) ? ucost : cost[r];
          mw[r] = own ? w : mw[r];
          okbits = own ? (RM)((okbits & ~(kOne << r)) | ((RM)(ulen < maxlen ? 1u : 0u) << r)) : okbits;
          fcpu[r] = (own && ufront) ? ucpu : fcpu[r];
          gn[r] = (own && ufront) ? ugn : gn[r];
#ifndef CNS_SEL_DIP_KEEP_ON_RELEASE   // (defined by a test build only: tests/test_preempt.py's stale-dip case must FAIL on it)
          if (kDip) dpa[r] = (own && released) ? kDipNone : dpa[r];
#endif
        }
    };
    // the dip the worker found under a rejected candidate (post_dip): the owner lane keeps it for the row
    auto take_dip = [&]() {
      if (!kDip) return;
      const u32 dc = uni32((u32)s_fl[4]);
      if (dc == kNone || owner_wave(dc) != wave) return;
      const u32 da = pack_dip((u32)s_fl[5], (u32)s_fl[6]), dg = (u32)s_fl[7];
      const int rr = (int)(dc >> 10);
      const bool own = (dc & 1023u) == t;
#pragma unroll
      for (int r = 0; r < kDipRows; ++r)
        if (r == rr) { dpa[r] = own ? da : dpa[r]; dpg[r] = own ? dg : dpg[r]; }
    };
    // One node that the pre-scan of job Sn skipped (a round-0 winner of the job before it): evaluate Sn's
    // filters on its refreshed registers and complete the pre-scanned candidate sets.  `code` is
    // wave-uniform and lies in this wave.
    auto fix_row = [&](const ScanJob& Sn, u64 tyok_n, u32 code) {
      const int rr = (int)(code >> 10);
      u32 w = 0, g = 0;
      int fc = 0;
#pragma unroll
      for (int r = 0; r < NPL; ++r)
        if (r == rr) { w = mw[r]; fc = fcpu[r]; g = gn[r]; }
      u32 gme = 0, gmo = 0;
      if (Sn.gmode & 2u) { gme = uni32(s_nme[Sn.gsel >> 8]); gmo = uni32(s_nmo[Sn.gsel >> 8]); }
      bool b, a;
      row_pred(Sn, tyok_n, gme, gmo, w, fc, g, b, a);
      const bool own = (code & 1023u) == t;
      bmask_n |= (RM)((own & b) ? 1u : 0u) << rr;
      amask_n |= (RM)((own & a) ? 1u : 0u) << rr;
    };

    u64 ji = jbeg;
    while (ji < jend) {
      PROF_T(s6);
      const bool excl_job = (J.flags & kJfExclusive) != 0;
      const u32 kk = J.k;
      const bool general = (J.shape & 1u) != 0;
      if (!pre_valid) {
        // ---- full scan of this job, publish both argmins ------------------------------------------------
        scan_job(J, ji, typeok, 0u, bmask, amask, ac, ap, tcs, tp);
        if (!idle) {
          wave_argmin(ac, ap);
          wave_argmin(tcs, tp);
        }
        if (lane == 0) { s_wc[par][wave] = ac; s_wp[par][wave] = ap; s_tc[wave] = tcs; s_tp[wave] = tp; }
        wg_barrier();  // B1
        wc = s_wc[par][lane & (kRed - 1)];
        wcode = s_wp[par][lane & (kRed - 1)];
        tc = s_tc[lane & (kRed - 1)];
        tcode = s_tp[lane & (kRed - 1)];
        reduce16(wc, wcode);
        reduce16(tc, tcode);
        par ^= 1;
      } else {
        // winners came from the worker's merge; the pre-scanned candidate sets were completed with the
        // skipped nodes right after the previous job's owner update (fix_row)
        bmask = bmask_n; amask = amask_n;
      }
      PROF_T(s2);
      PROF_ADDS(17, s6, s2);  // scanner: full scan + B1 (only when the pre-scan could not be used)

      // ---- while the worker examines the winner: decode the next job and PRE-SCAN it ----------------------
      // The only nodes this job can change (if it resolves in round 0, as ~99.9 % of single-node jobs do)
      // are the two round-0 winners.  The next job's candidate sets and argmins are computed now over all
      // OTHER nodes and published; after its commit the WORKER merges the winners' new state into them.
      wcode = uni32(wcode); tcode = uni32(tcode);
      ScanJob Jn = J;
      u64 typeok_n = typeok;
      const bool have_next = ji + 1 < jend;
      const bool shared_nodes = general_path_job(P.general_only != 0, P.sib_off != nullptr, J.flags, kk, general, (J.shape & 2u) != 0);
      const bool spec_ok = !excl_job && !general && kk == 1 && !shared_nodes && !P.sib_off;  // this job touches one node, a round-0 winner
      RM skipm = 0;
      if ((wcode & 1023u) == t && wcode != kNone) skipm |= kOne << (wcode >> 10);
      if ((tcode & 1023u) == t && tcode != kNone) {
        skipm |= kOne << (tcode >> 10);
        const int rr = (int)(tcode >> 10);
#pragma unroll
        for (int r = 0; r < NPL; ++r)
          if (r == rr) { s_on[0] = (u32)fcpu[r]; s_on[1] = mw[r]; s_on[2] = gn[r]; }
      }
      if (have_next) {
        Jn = decode(raw, typeok_n);
        if (ji + 2 < jend) raw = fetch_job(P, ji + 2);
        if (spec_ok) {
          u64 pc, ptc;
          u32 pp, ptp;
          scan_job(Jn, ji + 1, typeok_n, skipm, bmask_n, amask_n, pc, pp, ptc, ptp);
          if (!idle) {
            wave_argmin(pc, pp);
            wave_argmin(ptc, ptp);
          }
          if (lane == 0) { s_pc[wave] = pc; s_pp[wave] = pp; s_ptc[wave] = ptc; s_ptp[wave] = ptp; }
        }
      }
      PROF_T(s3);
      PROF_ADDS(18, s2, s3);  // scanner: next-job prep + pre-scan (hidden behind the worker)

      int verdict = 0;
      RM used = 0;
      bool round0 = true;  // resolved without a second scan round
      bool sequential = true;  // run the one-candidate-per-round protocol below
      if (!excl_job && !general && kk >= 2 && kk <= (u32)kMultiK && (J.shape & 2u) && !shared_nodes) {
        // ---- multi-node job, parallel protocol (mirror of the worker's): this wave lists its k best
        // candidates (sorted), the worker merges the lists, then wave i+1 verifies / commits candidate i ----
        sequential = false;
        round0 = false;
        auto post_topk = [&](RM mask) {
          RM m = mask;
          bool dry = idle;
          for (u32 i = 0; i < kk; ++i) {
            u64 c = ~0ull;
            u32 pc = kNone;
            if (!dry) {
              lane_argmin(m, c, pc);
              wave_argmin(c, pc);
            }
            if (lane == 0) { s_lc[(wave - 1) * kMultiK + i] = c; s_lp[(wave - 1) * kMultiK + i] = pc; }
            if (pc == kNone) dry = true;
            else if ((pc & 1023u) == t) m &= ~(kOne << (pc >> 10));
          }
        };
        post_topk(amask);
        wg_barrier();  // M1
        wg_barrier();  // M2: the worker merged the lists
        if (s_mode == 1) {
          if (multi_verify_commit(PG, &s_gres, &s_job, s_heap, wave - 1, wave - 1 < kk, qbeg, s_upd, &s_nupd)) {
            verdict = 2;
          } else {
            sequential = true;  // rare: fall back to the sequential protocol from round 0
          }
        } else {
          post_topk(bmask);
          wg_barrier();  // M5
          wg_barrier();  // M6
          if (s_mode == 3) {
            int reason = 0;
            const i64 st = multi_backfill_par(PG, &s_gres, &s_job, s_heap, wave - 1, wave - 1 < kk, qbeg, s_upd, &s_nupd, s_nf, &reason);
            if (st != kInf) verdict = 2;
          }
        }
      }
      // ---- Phase A ----------------------------------------------------------------------------------
      u32 acode = wcode;
      while (sequential && acode != kNone) {
        if ((acode & 1023u) == t) used |= kOne << (acode >> 10);
        wg_barrier();  // B2
        verdict = s_flag;
        if (verdict == 2) break;
        take_dip();
        round0 = false;
        lane_argmin(amask & ~used, ac, ap);
        wave_argmin(ac, ap);
        if (lane == 0) { s_wc[par][wave] = ac; s_wp[par][wave] = ap; }
        wg_barrier();  // B1
        u64 c2 = s_wc[par][lane & (kRed - 1)];
        acode = s_wp[par][lane & (kRed - 1)];
        reduce16(c2, acode);
        par ^= 1;
      }
      // ---- Phase B ----------------------------------------------------------------------------------
      if (sequential && verdict != 2) {
        if (!(!excl_job && !general && kk == 1)) {  // the single-node case needs no further scan
          used = 0;
          u32 nsel = 0;
          u32 ccode = tcode;
          while (ccode != kNone) {
            if ((ccode & 1023u) == t) used |= kOne << (ccode >> 10);
            if (!general) {
              if (++nsel == kk) break;
            } else {
              wg_barrier();  // B2 (ntasks > node_num only)
              if (s_flag == 1) break;
            }
            lane_argmin(bmask & ~used, tcs, tp);
            wave_argmin(tcs, tp);
            if (lane == 0) { s_wc[par][wave] = tcs; s_wp[par][wave] = tp; }
            wg_barrier();  // B1
            u64 cc = s_wc[par][lane & (kRed - 1)];
            ccode = s_wp[par][lane & (kRed - 1)];
            reduce16(cc, ccode);
            par ^= 1;
          }
        }
        wg_barrier();  // B3: worker finished backfill + commit (or gave up)
        verdict = s_flag;
      }
      PROF_T(s4);
      PROF_ADDS(19, s3, s4);  // scanner: waiting for the worker's verdict
      // a single-node job whose time map was longer than one chunk is committed out of line: rescan
      if (s_r0 == 0) round0 = false;  // the worker left its inline path (long time map): rescan
      // ---- owners refresh their registers ---------------------------------------------------------------
      if (verdict == 2) {
        const int nu = s_nupd;
        const UpdRec* const ub = (nu <= kMaxUpd && !shared_nodes && !P.sib_off) ? (const UpdRec*)s_upd : (const UpdRec*)(P.g_upd + qbeg);   // (a group that shares nodes: always the HBM list)
        for (int i = 0; i < nu; ++i) {
          const u32 up = uni32(ub[i].p);
          if (owner_wave(up) == wave) apply_upd(ub[i], up);  // only the owner's wave does any work
        }
      }
      pre_valid = have_next && spec_ok && round0 && !(Jn.flags & (kJfExclusive | kJfIncl | kJfExcl));
      if (pre_valid) {
        // complete the next job's pre-scanned sets with the nodes its pre-scan had to skip
        if (wcode != kNone && owner_wave(wcode) == wave) fix_row(Jn, typeok_n, wcode);
        if (tcode != kNone && tcode != wcode && owner_wave(tcode) == wave) fix_row(Jn, typeok_n, tcode);
      }
      PROF_T(s5);
      PROF_ADDS(20, s4, s5);  // scanner: owner update + completion of the pre-scanned sets
      if (pre_valid) {
        wg_barrier();  // B1': the worker merged the winners into the pre-scan
        wc = s_win_c[0]; wcode = s_win_p[0]; tc = s_win_c[1]; tcode = s_win_p[1];
      }
      J = Jn;
      typeok = typeok_n;
      PROF_T(s7);
      PROF_ADDS(21, s5, s7);  // scanner: wait for the worker's merge
      ++ji;
    }
  }
}

#undef s_flag
#include "pipe_kernel.inc"
// k_wide in four builds — scanner workgroups per partition / scanner waves / partitions of one launch it serves (every workgroup of
// the launch must be resident at once, the workgroups of a partition on one XCD): 16 / 64 / up to 8, 8 / 32 / up to 24,
// 4 / 16 / up to 48, 2 / 8 / up to 80.  The engine takes the widest one that fits the cluster (engine.hip: use_wide_kernel).
#ifndef CNS_WIDE_WGS
#define CNS_WIDE_WGS_ALL
#define CNS_WIDE_WGS 8
#endif
namespace w32 {
#include "wide_kernel.inc"
}
#ifdef CNS_WIDE_WGS_ALL
#undef CNS_WIDE_WGS
#define CNS_WIDE_WGS 16
namespace w64 {
#include "wide_kernel.inc"
}
#undef CNS_WIDE_WGS
#define CNS_WIDE_WGS 4
namespace w16 {
#include "wide_kernel.inc"
}
#undef CNS_WIDE_WGS
#define CNS_WIDE_WGS 2
namespace w8 {
#include "wide_kernel.inc"
}
#undef CNS_WIDE_WGS
#else
namespace w64 = w32;   // experiment builds with one explicit shape
namespace w16 = w32;
namespace w8 = w32;
#endif

#ifdef CNS_ONLY_NPL   // experiment builds: one tile width only
template __global__ void k_select<CNS_ONLY_NPL>(const KParams, const KParams*);
#else
#define CNS_INSTANTIATE(w) template __global__ void k_select<w>(const KParams, const KParams*);
CNS_NPL_LIST(CNS_INSTANTIATE)
#undef CNS_INSTANTIATE
#endif

}  // namespace cns
