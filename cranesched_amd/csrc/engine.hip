// Host side of the C ABI (include/crane_gpu/node_select.h): validation, SoA packing, HBM residency,
// kernel launches and timing.  No scheduling decision is taken on the host and there is no CPU
// fallback: without a usable HIP device cns_create fails with CNS_ERR_NO_DEVICE.
//
// Reference counterparts of the host-side packing (all in src/CraneCtld/JobScheduler.cpp):
//   split of the pending queue by partition            :6516-6530
//   partition -> node list, skip !alive || drain       :6569-6617
//   running jobs' per-node allocations                 :6681-6709
//   BasicPriority truncation to ScheduledBatchSize     JobScheduler.h:185-200
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/crane_gpu/node_select.h"
#include "../../include/crane_gpu/preempt.h"
#include "../../include/crane_gpu/priority.h"
#include "../../include/crane_gpu/run_limits.h"
#include "../../include/crane_gpu/steps.h"
#include <limits>
#include "engine_params.h"
#include <rccl/rccl.h>          // several devices: group_host.inc (the engine library links RCCL)
#include "select_kernels.hip"  // single translation unit: kernels + their launches (no -fgpu-rdc needed)
#include "priority_kernels.hip"
#include "limits_kernels.hip"
#include "steps_kernels.hip"
#include "jobs_host.inc"       // the host pass of cns_upload_jobs (no HIP in there: also compiled by the CPU tests)

using namespace cns;

namespace {

std::string g_create_error;

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap && p) return hipSuccess;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    size_t want = bytes ? bytes : 16;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() { if (p) { (void)hipFree(p); p = nullptr; cap = 0; } }
  template <class T> T* as() const { return static_cast<T*>(p); }
};

// Page-locked host staging for what cns_upload_jobs computes on the host (pre-set reasons, placement offsets, the grouped queue):
// a copy from a std::vector is a bounce through the runtime's own pinned buffer ON the calling thread; from here it is a DMA the
// thread does not wait for.  Grows by a quarter beyond the request (a queue that gains a few jobs per cycle re-pins nothing).
struct PinBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap && p) return hipSuccess;
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    const size_t want = bytes ? bytes + bytes / 4 : 64;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() { if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; } }
  template <class T> T* as() const { return static_cast<T*>(p); }
};

}  // namespace

struct cns_engine {
  cns_config cfg{};
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  hipStream_t stream2 = nullptr;                // second launch of a split cycle (the partitions that need k_select), beside the first
  hipEvent_t ev2[2] = {nullptr, nullptr};
  std::string err;

  // cluster (host copies)
  u32 N = 0, P = 0, S = 0, T = 0, max_np = 0;
  bool big_nodes = false;  // any GRES or > 64 cores (48-byte node record in the traffic model)
  std::vector<u32> part_off, slot_node, orig_pos_slot;  // orig_pos_slot: caller's part_nodes position -> slot or kNone
  std::vector<u32> node_slot;                 // node -> its PRIMARY slot (the one whose NodeBlock holds the shared time map)
  // Overlapping partitions (one NodeState per craned shared by every partition that lists it, one cost per partition:
  // JobScheduler.cpp:6585-6615, JobScheduler.h:498-516): partitions connected through shared nodes form ONE engine
  // partition (workgroup) that runs their jobs in queue order; a node then has one slot per member partition.
  u32 Pu = 0;                                 // partitions of the caller
  bool shared = false;                        // some node belongs to several partitions
  u32 num_cus = 0;                            // compute units of the device (0: unknown); k_wide needs one per workgroup, all resident at once
  std::vector<u32> upart_eng, upart_size;     // caller's partition -> engine partition, its schedulable node count
  std::vector<uint8_t> refused_probe;         // the statuses of the last cns_set_nodes call, also when it failed because EVERY partition was refused
  std::vector<uint8_t> upart_refused;         // caller's partition -> cns_partition_status: != 0: its group is outside the engine's limits — its jobs get
                                              // CNS_REASON_ENGINE_REFUSED, every other partition is served (cns_get_partition_status)
  std::vector<uint8_t> upart_tag;             // ... and its member tag inside that engine partition
  std::vector<std::vector<u32>> node_slots;   // node -> all its slots
  std::vector<uint8_t> slot_tag;              // slot -> member tag
  DevBuf d_slot_block, d_sib_off, d_sib, d_type_tag, d_jtag;
  GresDev gres{};
  bool have_nodes = false, have_jobs = false, have_run = false;

  // layout: slots [0, S_real) are the partitions' nodes, slots [S_real, S) the virtual nodes of the reservations
  // (one extra "partition" P_real + v per reservation v, JobScheduler.cpp:6657-6668)
  u32 P_real = 0, S_real = 0, V = 0;
  std::vector<Res> node_total;                       // res_total per node (host copy)
  std::vector<Res> slot_total;                       // res_total per slot (virtual slots: the reserved share)
  std::vector<i64> slot_end;                         // end of the slot's time map (INF, or the reservation's end)
  std::vector<i64> resv_start, resv_end;             // per reservation
  std::vector<std::map<u32, u32>> resv_node_slot;    // per reservation: node -> slot
  std::vector<u32> rv_off;                           // [S+1] reservation entries touching a REAL slot
  std::vector<i64> rv_start, rv_endt;
  std::vector<Res> rv_res;
  DevBuf d_slot_total, d_slot_end, d_slot_type, d_rv_off, d_rv_start, d_rv_end, d_rv_res, d_first_resv, d_resv_se;
  // device buffers
  DevBuf d_part_off, d_slot_node, d_type_total, d_blocks, d_cost, d_fcpu,
      d_fmem, d_fcnt, d_dipt, d_dipcm, d_dipg, d_rn_off, d_rn_end, d_rn_res, d_heap, d_bfj, d_gupd, d_fault;
  DevBuf d_pj_off, d_jobs, d_incl, d_excl, d_reason_init, d_results, d_params, d_prof, d_wide;
  DevBuf d_raw[16];  // the caller's job arrays as uploaded (k_pack_jobs reads them; d_raw[14] = place offsets)
  // job table
  u64 J = 0, Jg = 0, places = 0, jobs_ordered = 0, algo_bytes = 0;
  u64 window_shaped = 0;   // jobs of the uploaded queue that a window of k_wide can decide: one node, one task per node, no GRES, no node lists, not exclusive
  PinBuf h_place, h_grouped, h_reason, h_jtag;   // host staging of cns_upload_jobs (page-locked)
  u32 host_threads = 0;                         // cns_set_host_threads (0: CNS_HOST_THREADS, else up to 16)
  const u64* place_off = nullptr;               // [J + 1] first placement record per job (in h_place)
  struct ResOff { size_t start, cpu, mem, clo, chi, gres, node, ntasks, reason, c2, c3, total; } ro{};
  bool wide_cores = false;   // a node of the snapshot has a core id above 127: the results carry the core_w2 / core_w3 planes
  cns_timing timing{};
  std::string last_kernel;
  i64 last_now = 0;
  std::vector<u32> eng_members;                 // engine partition -> number of caller partitions it runs (> 1: they share nodes)
  std::vector<u32> job_part;                    // pending job (queue index) -> engine partition (kNone: not given to the ordered loop)
  std::vector<u64> part_jobs;                   // engine partition -> jobs of the uploaded queue that reach its ordered loop
  std::vector<uint8_t> pre_part;                // cycle with preemption: engine partition has a pending job whose qos may preempt
  DevBuf d_params2, d_pmap_a, d_pmap_b, d_wide_last;
  DevBuf d_params3, d_pmap_c, d_wide_mem;       // the serial-only launch of k_wide (groups wider than k_select's tile)
  DevBuf d_flen, d_tag_off, d_tag_base;         // ... its compact map lengths, and the slot range of every member partition of a group
  std::vector<u32> tag_off, tag_base;
  // several devices (group_host.inc): this engine's rank in a communicator, the all-gathered results of every rank
  void* comm = nullptr;                         // ncclComm_t
  u32 comm_nranks = 0, comm_rank = 0;
  DevBuf d_gather;
  double gather_ms = 0.0;
  u64 gather_bytes = 0;
  bool wide_off = false;                        // this run must not use k_wide (the retry after a k_wide protocol fault)
  u32 wide_retries = 0;                         // cycles that were re-run on k_pipe / k_select after a k_wide fault (lifetime of the handle)
  // MultiFactorPriority (priority_host.inc)
  DevBuf d_prio[27];
  double prio_ms = 0.0;
  u64 prio_bytes = 0;
  // run-limit admission (limits_host.inc)
  DevBuf d_lim[29], d_limpar[17];
  DevBuf d_step[13];  // step scheduler (steps_host.inc)
  // preemption (include/crane_gpu/preempt.h): what cns_set_running kept of the running set, and the cycle's tables
  u32 R = 0;                                    // running jobs of the last cns_set_running
  std::vector<u32> ent_job, ent_slot;           // slot-grouped allocation entry d -> running job, slot
  std::vector<i64> ent_end;                     // ... -> end time as handed in
  std::set<void*> host_bufs;                    // page-locked host buffers handed out by cns_host_alloc
  bool pre_active = false;                      // the next run is a cycle with preemption (general path of k_select only)
  PreParams pre_params{};
  DevBuf d_pre[24];
  bool lim_have_tables = false, lim_have_jobs = false, lim_have_run = false;
  bool lim_has_upl = false, lim_has_apl = false, lim_has_sel = false, lim_has_skip = false;
  u32 lim_U = 0, lim_UA = 0, lim_A = 0, lim_Q = 0, lim_Pn = 0, lim_base[5] = {0, 0, 0, 0, 0};
  u64 lim_NR = 0, lim_J = 0, lim_sel_J = 0;
  std::vector<u32> lim_level;
  cns_limit_timing lim_timing{};
};

namespace {

int fail(cns_engine* h, int code, const std::string& msg) {
  if (h) h->err = msg; else g_create_error = msg;
  return code;
}
#define HIPCHK(h, call)                                                                         \
  do {                                                                                          \
    hipError_t _e = (call);                                                                     \
    if (_e != hipSuccess)                                                                       \
      return fail(h, CNS_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_e));           \
  } while (0)

template <class T>
int upload(cns_engine* h, DevBuf& b, const std::vector<T>& v) {
  HIPCHK(h, b.ensure(v.size() * sizeof(T)));
  if (!v.empty()) HIPCHK(h, hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, h->stream));
  return 0;
}

size_t align16(size_t x) { return (x + 15) & ~size_t(15); }

int build_gres(cns_engine* h, const cns_gres_layout& g) {
  GresDev d{};
  if (g.num_classes > CNS_MAX_GRES_CLASSES) return fail(h, CNS_ERR_INVALID_ARG, "gres.num_classes > 8");
  d.num_classes = g.num_classes;
  u64 seen = 0;
  for (u32 c = 0; c < g.num_classes; ++c) {
    if (g.class_name[c] >= CNS_MAX_GRES_NAMES) return fail(h, CNS_ERR_INVALID_ARG, "gres class name id >= 4");
    if (g.class_width[c] == 0 || g.class_shift[c] + g.class_width[c] > 64)
      return fail(h, CNS_ERR_INVALID_ARG, "gres class bit range outside 64-bit mask");
    u64 w = g.class_width[c] >= 64 ? ~0ull : ((1ull << g.class_width[c]) - 1ull);
    u64 m = w << g.class_shift[c];
    if (seen & m) return fail(h, CNS_ERR_INVALID_ARG, "gres classes overlap");
    seen |= m;
    d.class_mask[c] = m;
    d.class_name_packed |= (u32)g.class_name[c] << (4 * c);
    d.name_mask[g.class_name[c]] |= m;
    d.name_bytes[g.class_name[c]] |= 0xFFull << (8 * c);
  }
  h->gres = d;
  return 0;
}

void fill_params(cns_engine* h, KParams& K, i64 now) {
  memset(&K, 0, sizeof K);
  K.num_nodes = h->N; K.num_parts = h->P; K.num_slots = h->S; K.num_types = h->T;
  K.tl_cap = kTlCap;
  K.wide_cores = h->wide_cores ? 1u : 0u;
  K.max_jobs_per_node = h->cfg.max_job_num_per_node;
  K.now = now;
  K.max_window = h->cfg.max_time_window_sec;
  if (const char* inj = getenv("CNS_WIDE_INJECT_STALL")) K.wide_inject_stall = (u32)strtoul(inj, nullptr, 10) + 1u;
  // jobs per pool exchange of k_wide's 64-wave build at most (wide_kernel.inc, "A WINDOW OF JOBS PER EXCHANGE"); 0 / 1: one job per exchange, as in
  // rounds 2-4 — and the kernel WITHOUT the window path is launched (k_wide<NPL, false>: the path's presence costs the single-job loops 8 %).
  // Unless CNS_WIDE_WINDOW says otherwise the QUEUE decides: windows for a queue that is (almost) all one-node jobs without GRES and node lists
  // (>= 95 %: C5 190 -> 185 ms, C2 / c5deep unchanged — where most jobs are backfilled the windows back off), none for a mix like C4's,
  // whose GRES backfills and multi-node jobs would close every window after two or three jobs (258 against 273 ms):
  // profiles/r05_ab_window_code_presence.txt, DESIGN.md 5.8.
  K.wide_window = (h->jobs_ordered != 0 && h->window_shaped * 100 >= h->jobs_ordered * 95) ? w64::kWJ : 0u;
  if (const char* ww = getenv("CNS_WIDE_WINDOW")) { const u32 v = (u32)strtoul(ww, nullptr, 10); K.wide_window = v < w64::kWJ ? v : w64::kWJ; }
  if (K.wide_inject_stall) K.wide_window = 0;
  K.wide_tester_opt = 1;
  if (const char* to = getenv("CNS_WIDE_TESTER_OPT")) K.wide_tester_opt = (u32)strtoul(to, nullptr, 10);
  K.wide_aux = 0;   // (sized per launch: launch_wide)
  K.wide_batch_post = 1;
  if (const char* bp = getenv("CNS_WIDE_BATCH_POST")) K.wide_batch_post = (u32)strtoul(bp, nullptr, 10);
  K.part_off = h->d_part_off.as<u32>();
  K.slot_node = h->d_slot_node.as<u32>();
  K.slot_total = h->d_slot_total.as<Res>();
  K.slot_end = h->d_slot_end.as<i64>();
  K.slot_type = h->d_slot_type.as<uint8_t>();
  K.rv_off = h->d_rv_off.as<u32>();
  K.rv_start = h->d_rv_start.as<i64>();
  K.rv_end = h->d_rv_end.as<i64>();
  K.rv_res = h->d_rv_res.as<Res>();
  K.first_resv = h->d_first_resv.as<i64>();
  K.resv_se = h->d_resv_se.as<i64>();
  K.num_real_parts = h->P_real;
  K.type_total = h->d_type_total.as<Res>();
  K.blocks = h->d_blocks.as<char>();
  K.block_stride = kBlockStride;
  K.cost = h->d_cost.as<double>();
  K.f_cpu = h->d_fcpu.as<int>();
  K.f_mem = h->d_fmem.as<u32>();
  K.f_cnt = h->d_fcnt.as<u64>();
  K.dip_t = h->d_dipt.as<u32>(); K.dip_cm = h->d_dipcm.as<u32>(); K.dip_g = h->d_dipg.as<u32>();
  if (h->shared) { K.f_len = h->d_flen.as<u32>(); K.tag_off = h->d_tag_off.as<u32>(); K.tag_base = h->d_tag_base.as<u32>(); }
  K.rn_off = h->d_rn_off.as<u32>();
  K.rn_end = h->d_rn_end.as<i64>();
  K.rn_res = h->d_rn_res.as<Res>();
  K.pj_off = h->d_pj_off.as<u64>();
  K.jobrec = h->d_jobs.as<u32>();
  K.incl_nodes = h->d_incl.as<u32>();
  K.excl_nodes = h->d_excl.as<u32>();
  char* rb = h->d_results.as<char>();
  K.o_start = (i64*)(rb + h->ro.start);
  K.o_cpu = (i64*)(rb + h->ro.cpu);
  K.o_mem = (u64*)(rb + h->ro.mem);
  K.o_clo = (u64*)(rb + h->ro.clo);
  K.o_chi = (u64*)(rb + h->ro.chi);
  K.o_c2 = h->wide_cores ? (u64*)(rb + h->ro.c2) : nullptr;
  K.o_c3 = h->wide_cores ? (u64*)(rb + h->ro.c3) : nullptr;
  K.o_gres = (u64*)(rb + h->ro.gres);
  K.o_node = (u32*)(rb + h->ro.node);
  K.o_ntasks = (u32*)(rb + h->ro.ntasks);
  K.o_reason = (uint8_t*)(rb + h->ro.reason);
  K.heap = h->d_heap.as<HeapEnt>();
  K.bf_j = h->d_bfj.as<u32>();
  K.g_upd = h->d_gupd.as<UpdRec>();
  K.fault = h->d_fault.as<u32>();
  K.prof = h->d_prof.as<u64>();
  K.wide_ctl = h->d_wide.as<char>();
  K.general_only = h->pre_active ? 1u : 0u;
  K.serial_only = 0;
  K.pre = h->pre_active ? h->pre_params : PreParams{};
  K.gres = h->gres;
  if (h->shared) {
    K.slot_block = h->d_slot_block.as<u32>(); K.sib_off = h->d_sib_off.as<u32>(); K.sib = h->d_sib.as<u32>();
    K.slot_tag = h->d_type_tag.as<uint8_t>();
  }
}

// One launch of a cycle: the partitions it serves (all of them, or the part_map of K), its stream and its copy of the
// parameter block in HBM (the out-of-line routines read that one).
struct LaunchCtx {
  u32 nparts;            // partitions of this launch
  u32 max_np;            // widest of them (slots)
  hipStream_t stream;
  const KParams* dparams;
  u32 other_blocks;      // workgroups of the cycle's other launch (they hold CUs while k_wide needs all of its own resident)
};
template <int NPL>
void launch_select(cns_engine* h, const KParams& K, const LaunchCtx& L) {
  hipLaunchKernelGGL((k_select<NPL>), dim3(L.nparts), dim3(kBlock), 0, L.stream, K, L.dparams);
}
template <int NPL>
void launch_pipe(cns_engine* h, const KParams& K, const LaunchCtx& L) {
  hipLaunchKernelGGL((k_pipe<NPL>), dim3(L.nparts), dim3(kPBlock), 0, L.stream, K, L.dparams);
}
// k_wide: 1 + 8 (or 1 + 16) workgroups per partition; the workgroups of a partition share blockIdx % 8 (= the XCD, observed).
// A pad of dynamic LDS keeps it at one workgroup per CU (one scanner wave per SIMD is the point of the kernel).
template <class W>
int launch_wide(cns_engine* h, const KParams& K, const LaunchCtx& L, std::string* name) {
  const u32 np = L.max_np;
  const char* kname = "";
  const void* fn = W::pick(np, &kname, K.wide_window >= 2u);
  if (!fn) return 2;
  const unsigned groups = (L.nparts + 7u) / 8u;
  // Extra home workgroups per partition (wide_kernel.inc, "MORE THAN ONE HOME WORKGROUP PER PARTITION"): as many as the build allows, as long as
  // every workgroup of the launch still gets a CU of "its" XCD (32 each, `groups` partitions per XCD) and of the device (other launches of the
  // cycle hold theirs); none for tiles whose last-task table is not in LDS.  ONE extra home is the default: with two homes the scanners pace every
  // configuration measured (C5 184.6 -> 146.6 ms, C2 139.4 -> 114.2; a third and a fourth home: 146.3 / 114.2 — profiles/r06_ab_home_workgroups.txt).
  // CNS_WIDE_AUX=<n> sets the number (0: rounds 2-5's single home; up to the build's maximum: the parity tests run them all).
  unsigned aux = np > W::lanes * W::last_in_lds_rows ? 0u : (W::aux_max < 1u ? W::aux_max : 1u);
  if (const char* ea = getenv("CNS_WIDE_AUX")) { const unsigned v = (unsigned)strtoul(ea, nullptr, 10); aux = np > W::lanes * W::last_in_lds_rows ? 0u : (v < W::aux_max ? v : W::aux_max); }
  while (aux > 0 && (groups * (W::group + aux) > 32u || (u64)8u * groups * (W::group + aux) + L.other_blocks > h->num_cus)) --aux;
  const unsigned grid = 8u * groups * (W::group + aux);
  const size_t need = (size_t)h->P * W::ctl_bytes;
  if (h->d_wide.ensure(need) != hipSuccess) return 1;
  if (hipMemsetAsync(h->d_wide.p, 0, need, L.stream) != hipSuccess) return 1;
  KParams K2 = K;
  K2.wide_ctl = h->d_wide.as<char>();
  K2.wide_aux = aux;
  if (np > W::lanes * 4u) {   // 8 / 16 rows per lane: the home workgroup's last-task table does not fit the LDS
    const size_t lb = (size_t)h->P * W::lanes * W::npl_max * sizeof(u32);
    if (h->d_wide_last.ensure(lb) != hipSuccess) return 1;
    if (hipMemsetAsync(h->d_wide_last.p, 0, lb, L.stream) != hipSuccess) return 1;
    K2.wide_last = h->d_wide_last.as<u32>();
  }
  if (hipMemcpyAsync(&const_cast<KParams*>(L.dparams)->wide_ctl, &K2.wide_ctl, sizeof(char*), hipMemcpyHostToDevice, L.stream) != hipSuccess) return 1;
  size_t dyn = 0;
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, fn) == hipSuccess && fa.sharedSizeBytes < 84u * 1024u) {
    dyn = 84u * 1024u - fa.sharedSizeBytes;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) { dyn = 0; (void)hipGetLastError(); }
  }
  // The workgroups of this launch spin on each other: ALL of them must be resident at once.  Proof, not assumption: the
  // runtime's own occupancy figure for this kernel at this block size and LDS footprint, times the device's CUs, must
  // cover the grid — else the launch is refused here and the cycle runs on k_pipe / k_select, which need no co-residency.
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, (int)W::block, dyn) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
  if (per_cu < 1 || (u64)per_cu * h->num_cus < (u64)grid + L.other_blocks) return 2;
  const KParams* dparams = L.dparams;
  void* args[2] = {(void*)&K2, (void*)&dparams};
  if (hipLaunchKernel(fn, dim3(grid), dim3(W::block), args, dyn, L.stream) != hipSuccess) return 1;
  *name = std::string(kname) + " x" + std::to_string(W::waves);   // (x64: cns::w64::k_wide in a profile, x32: cns::w32::k_wide, ...)
  return 0;
}
// Groups of partitions that share nodes and are wider than k_select's register tile (an "ALL" partition over a large cluster):
// k_wide's HOME workgroup alone, every job through the sequential protocol with its tester waves as memory scanners over the
// committed HBM arrays (KParams::serial_only).  The narrowest build serves (its two scanner workgroups per partition leave at once);
// no workgroup waits for another one, so no co-residency is needed.  Slow — every job reads every slot of its group — and exact.
int launch_mem(cns_engine* h, const KParams& K, const LaunchCtx& L, std::string* name) {
  using W = w8::WideInfo;
  const void* fn = (const void*)w8::k_wide<1, false>;
  const unsigned groups = (L.nparts + 7u) / 8u;
  const unsigned grid = 8u * groups * W::group;
  const size_t need = (size_t)h->P * W::ctl_bytes;
  if (h->d_wide_mem.ensure(need) != hipSuccess) return 1;
  if (hipMemsetAsync(h->d_wide_mem.p, 0, need, L.stream) != hipSuccess) return 1;
  KParams K2 = K;
  K2.wide_ctl = h->d_wide_mem.as<char>();
  K2.serial_only = 1;
  if (hipMemcpyAsync(const_cast<KParams*>(L.dparams), &K2, sizeof(KParams), hipMemcpyHostToDevice, L.stream) != hipSuccess) return 1;
  size_t dyn = 0;
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, fn) == hipSuccess && fa.sharedSizeBytes < 84u * 1024u) {
    dyn = 84u * 1024u - fa.sharedSizeBytes;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) { dyn = 0; (void)hipGetLastError(); }
  }
  const KParams* dparams = L.dparams;
  void* args[2] = {(void*)&K2, (void*)&dparams};
  if (hipLaunchKernel(fn, dim3(grid), dim3(W::block), args, dyn, L.stream) != hipSuccess) return 1;
  *name = "k_mem (k_wide<1> home workgroup, sequential protocol over the HBM arrays)";
  return 0;
}
// Which selection kernel runs: k_pipe (decoupled test / commit pipeline) for partitions its tile covers, k_select
// otherwise.  CNS_SELECT_KERNEL=legacy|pipe forces one (A/B measurements, and the parity tests run both).
#ifndef CNS_DEFAULT_PIPE
#define CNS_DEFAULT_PIPE 1
#endif
#ifndef CNS_DEFAULT_WIDE
#define CNS_DEFAULT_WIDE 1
#endif
// k_wide (many CUs per partition) when every workgroup of the launch can be resident at once and the partitions fit its tile:
// 64 scanner waves per partition (17 workgroups) for up to 8 partitions, 32 (9 workgroups) for up to 24, 16 (5) for up to 48,
// 8 (3) for up to 80 — the widest build that fits.  Returns 0 / 8 / 16 / 32 / 64.
// CNS_SELECT_KERNEL=wide32 | wide16 | wide8 caps the build (A/B measurements, and the parity tests run them).
// (for a launch over partitions that neither share nodes nor run with preemption: the others go to k_select)
u32 use_wide_kernel(const cns_engine* h, const LaunchCtx& L) {
  const char* e = getenv("CNS_SELECT_KERNEL");
  bool want = CNS_DEFAULT_WIDE != 0;
  if (e && (!strcmp(e, "legacy") || !strcmp(e, "pipe"))) want = false;
  u32 cap = 64;
  if (e && !strcmp(e, "wide32")) cap = 32;
  if (e && !strcmp(e, "wide16")) cap = 16;
  if (e && !strcmp(e, "wide8")) cap = 8;
  if (e && (!strcmp(e, "wide") || cap != 64)) want = true;
  if (!e && h->cfg.kernel_pin != CNS_KERNEL_AUTO) want = false;   // (cns_config::kernel_pin: a controller that shares its GPU; the environment variable, an A/B and test switch, wins)
  if (!want || h->wide_off) return 0;
  // every workgroup of the launch must be resident at once, one per CU (a partitioned or smaller device falls to k_pipe)
  const u32 groups = (L.nparts + 7u) / 8u;
  auto fits = [&](u32 wgs_per_part) { return h->num_cus != 0 && 8u * groups * wgs_per_part + L.other_blocks <= h->num_cus; };   // (unknown CU count: no proof of co-residency, no k_wide)
  auto serves = [&](u32 waves, u32 max_parts, u32 slots, u32 group) { return cap >= waves && L.nparts <= max_parts && L.max_np <= slots && fits(group); };
  if (serves(64, w64::WideInfo::max_parts, w64::WideInfo::lanes * w64::WideInfo::npl_max, w64::WideInfo::group)) return 64;
  if (serves(32, w32::WideInfo::max_parts, w32::WideInfo::lanes * w32::WideInfo::npl_max, w32::WideInfo::group)) return 32;
  if (serves(16, w16::WideInfo::max_parts, w16::WideInfo::lanes * w16::WideInfo::npl_max, w16::WideInfo::group)) return 16;
  if (serves(8, w8::WideInfo::max_parts, w8::WideInfo::lanes * w8::WideInfo::npl_max, w8::WideInfo::group)) return 8;
  return 0;
}
bool use_pipe_kernel(const cns_engine* h, const LaunchCtx& L) {
  const char* e = getenv("CNS_SELECT_KERNEL");
  bool want = CNS_DEFAULT_PIPE != 0;
  if (e && !strcmp(e, "legacy")) want = false;
  if (e && (!strcmp(e, "pipe") || !strncmp(e, "wide", 4))) want = true;
  if (!e && h->cfg.kernel_pin == CNS_KERNEL_SELECT) want = false;
  return want && L.max_np <= kPScan * (u32)kPNplMax;
}
// One launch: k_wide / k_pipe where the partitions allow it (`plain`: none of them shares nodes or runs with preemption), else
// k_select.  0 ok (kernel name in *name), else a status for fail().
int launch_one(cns_engine* h, const KParams& K, const LaunchCtx& L, bool plain, std::string* name, std::string* err) {
  const u32 np = L.max_np;
#ifdef CNS_ONLY_NPL   // experiment builds: one tile width only
  if (np > kScan * CNS_ONLY_NPL) { *err = "experiment build: partition too large for its one tile width"; return CNS_ERR_UNSUPPORTED; }
  launch_select<CNS_ONLY_NPL>(h, K, L);
  *name = "k_select";
  return 0;
#else
  bool launched = false;
  if (const u32 ww = plain ? use_wide_kernel(h, L) : 0u) {
    const int rc = ww == 64 ? launch_wide<w64::WideInfo>(h, K, L, name) : ww == 32 ? launch_wide<w32::WideInfo>(h, K, L, name)
                 : ww == 16 ? launch_wide<w16::WideInfo>(h, K, L, name) : launch_wide<w8::WideInfo>(h, K, L, name);
    if (rc == 1) { *err = "k_wide: control block allocation / upload / launch failed"; return CNS_ERR_HIP; }
    launched = rc == 0;
  }
  if (!launched && plain && use_pipe_kernel(h, L)) {
#define CNS_TRY_PWIDTH(w) if (!launched && np <= kPScan * (w)) { launch_pipe<w>(h, K, L); launched = true; *name = "k_pipe<" #w ">"; }
    CNS_PNPL_LIST(CNS_TRY_PWIDTH)
#undef CNS_TRY_PWIDTH
  }
#define CNS_TRY_WIDTH(w) if (!launched && np <= kScan * (w)) { launch_select<w>(h, K, L); launched = true; *name = "k_select<" #w ">"; }
  CNS_NPL_LIST(CNS_TRY_WIDTH)
#undef CNS_TRY_WIDTH
  if (!launched) { *err = "partition too large for the widest register tile"; return CNS_ERR_UNSUPPORTED; }
  return 0;
#endif
}

// Everything that depends on the slot list (real + virtual): per-slot res_total / time-map end, node types
// (= distinct res_total records), the device copies and the per-slot buffers.
int finalize_layout(cns_engine* h, const std::vector<Res>* virt_total = nullptr) {
  const u32 S = h->S;
  h->slot_total.resize(S);
  h->slot_end.assign(S, INT64_MAX);
  for (u32 q = 0; q < h->S_real; ++q) h->slot_total[q] = h->node_total[h->slot_node[q]];
  for (u32 q = h->S_real; q < S; ++q) h->slot_total[q] = (*virt_total)[q - h->S_real];
  for (u32 v = 0; v < h->V; ++v)
    for (u32 q = h->part_off[h->P_real + v]; q < h->part_off[h->P_real + v + 1]; ++q) h->slot_end[q] = h->resv_end[v];
  // (per partition / group: checked in cns_set_nodes — a partition that shares no node may be as wide as k_wide's widest tile)
  if (h->max_np > std::max<u32>(w8::WideInfo::mem_slots, w64::WideInfo::lanes * w64::WideInfo::npl_max))
    return fail(h, CNS_ERR_UNSUPPORTED, "partition with more than " + std::to_string(std::max<u32>(w8::WideInfo::mem_slots, w64::WideInfo::lanes * w64::WideInfo::npl_max)) + " schedulable (partition, node) slots");
  std::map<std::tuple<i64, u64, u64, u64, u64, u64, u64>, u32> tmap;
  std::vector<Res> type_total;
  std::vector<uint8_t> slot_type(std::max<u32>(S, 1), 0);
  h->slot_tag.resize(S, 0);   // virtual (reservation) slots: tag 0
  for (u32 q = 0; q < S; ++q) {
    const Res& r = h->slot_total[q];
    auto key = std::make_tuple(r.cpu, r.mem, r.clo, r.chi, r.gres, r.c2, r.c3);
    auto it = tmap.find(key);
    if (it == tmap.end()) {
      if (type_total.size() >= CNS_MAX_NODE_TYPES)
        return fail(h, CNS_ERR_UNSUPPORTED, "more than 64 distinct res_total records (nodes + reservation shares)");
      it = tmap.emplace(key, (u32)type_total.size()).first;
      type_total.push_back(r);
    }
    slot_type[q] = (uint8_t)it->second;
  }
  h->T = (u32)type_total.size();
  if (h->shared) {
    std::vector<u32> slot_block(S), sib_off(S + 1, 0), sib;
    for (u32 q = 0; q < S; ++q) {
      slot_block[q] = q;
      if (q < h->S_real) {
        const auto& all = h->node_slots[h->slot_node[q]];
        slot_block[q] = all.front();
        for (u32 o : all) if (o != q) sib.push_back(o);
      }
      sib_off[q + 1] = (u32)sib.size();
    }
    if (sib.empty()) sib.push_back(0);
    if (int rc = upload(h, h->d_slot_block, slot_block)) return rc;
    if (int rc = upload(h, h->d_sib_off, sib_off)) return rc;
    if (int rc = upload(h, h->d_sib, sib)) return rc;
    if (int rc = upload(h, h->d_type_tag, h->slot_tag)) return rc;   // per slot: member partition inside the group
  }
  std::vector<i64> resv_se(2 * std::max<u32>(h->V, 1), 0);
  for (u32 v = 0; v < h->V; ++v) { resv_se[2 * v] = h->resv_start[v]; resv_se[2 * v + 1] = h->resv_end[v]; }
  if (int rc = upload(h, h->d_part_off, h->part_off)) return rc;
  if (int rc = upload(h, h->d_slot_node, h->slot_node)) return rc;
  if (int rc = upload(h, h->d_slot_total, h->slot_total)) return rc;
  if (int rc = upload(h, h->d_slot_end, h->slot_end)) return rc;
  if (int rc = upload(h, h->d_slot_type, slot_type)) return rc;
  if (int rc = upload(h, h->d_type_total, type_total)) return rc;
  if (int rc = upload(h, h->d_rv_off, h->rv_off)) return rc;
  {
    std::vector<i64> a = h->rv_start, b = h->rv_endt;
    std::vector<Res> c = h->rv_res;
    if (a.empty()) { a.push_back(0); b.push_back(0); c.push_back(Res{0, 0, 0, 0, 0}); }
    if (int rc = upload(h, h->d_rv_start, a)) return rc;
    if (int rc = upload(h, h->d_rv_end, b)) return rc;
    if (int rc = upload(h, h->d_rv_res, c)) return rc;
  }
  if (int rc = upload(h, h->d_resv_se, resv_se)) return rc;
  const size_t S1 = std::max<u32>(S, 1);
  HIPCHK(h, h->d_blocks.ensure(S1 * kBlockStride));  // 64.6 KB per node: HBM is plentiful
  HIPCHK(h, h->d_cost.ensure(S1 * sizeof(double)));
  HIPCHK(h, h->d_fcpu.ensure(S1 * sizeof(int)));
  HIPCHK(h, h->d_fmem.ensure(S1 * sizeof(u32)));
  HIPCHK(h, h->d_fcnt.ensure(S1 * sizeof(u64)));
  HIPCHK(h, h->d_dipt.ensure(S1 * sizeof(u32))); HIPCHK(h, h->d_dipcm.ensure(S1 * sizeof(u32))); HIPCHK(h, h->d_dipg.ensure(S1 * sizeof(u32)));
  if (h->shared) {
    HIPCHK(h, h->d_flen.ensure(S1 * sizeof(u32)));
    std::vector<u32> tb = h->tag_base;
    tb.resize(std::max<size_t>(h->P, 1), 0);   // (the virtual partitions of reservations share nothing: never read)
    if (int rc = upload(h, h->d_tag_base, tb)) return rc;
    if (int rc = upload(h, h->d_tag_off, h->tag_off)) return rc;
  }
  HIPCHK(h, h->d_first_resv.ensure(S1 * sizeof(i64)));
  HIPCHK(h, h->d_heap.ensure((size_t)(S + h->P + 1) * sizeof(HeapEnt)));
  HIPCHK(h, h->d_bfj.ensure(S1 * sizeof(u32)));
  HIPCHK(h, h->d_gupd.ensure(S1 * sizeof(UpdRec)));
  HIPCHK(h, h->d_fault.ensure(4 * sizeof(u32)));
  HIPCHK(h, h->d_prof.ensure(((size_t)(h->P + 8) * (size_t)w64::WideInfo::group * 32 + (size_t)h->P * 8 + 2048) * sizeof(u64)));   // (k_wide: blocks > partitions)
  // no running jobs until cns_set_running
  std::vector<u32> rn_off(S + 1, 0);
  if (int rc = upload(h, h->d_rn_off, rn_off)) return rc;
  HIPCHK(h, h->d_rn_end.ensure(16));
  HIPCHK(h, h->d_rn_res.ensure(sizeof(Res)));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}

}  // namespace

extern "C" {

int cns_abi_version(void) { return (int)CNS_ABI_VERSION; }

const char* cns_last_error(const cns_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int cns_create(const cns_config* cfg, cns_handle** out) {
  if (!cfg || !out) return fail(nullptr, CNS_ERR_INVALID_ARG, "cns_create: null argument");
  if (cfg->abi_version != CNS_ABI_VERSION) return fail(nullptr, CNS_ERR_INVALID_ARG, "cns_create: ABI version mismatch");
  if (cfg->kernel_pin > CNS_KERNEL_PIPE) return fail(nullptr, CNS_ERR_INVALID_ARG, "cns_create: kernel_pin is not a cns_kernel_pin (a field the caller never zeroed?)");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(nullptr, CNS_ERR_NO_DEVICE, std::string("no HIP device: ") + hipGetErrorString(e) +
                                                " (the engine has no CPU fallback)");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, CNS_ERR_INVALID_ARG, "cns_create: bad device ordinal");
  cns_engine* h = new (std::nothrow) cns_engine();
  if (!h) return fail(nullptr, CNS_ERR_HIP, "out of host memory");
  h->cfg = *cfg;
  if (h->cfg.max_job_num_per_node == 0) h->cfg.max_job_num_per_node = 1000;  // kAlgoMaxJobNumPerNode
  if (h->cfg.max_time_window_sec == 0) h->cfg.max_time_window_sec = 7 * 24 * 3600;  // kAlgoMaxTimeWindow
  if (h->cfg.max_job_num_per_node + 2 > kTlCap) {
    delete h;
    return fail(nullptr, CNS_ERR_UNSUPPORTED, "max_job_num_per_node > 1006");
  }
  h->device = cfg->device;
  if ((e = hipSetDevice(h->device)) != hipSuccess || (e = hipStreamCreate(&h->stream)) != hipSuccess) {
    std::string m = hipGetErrorString(e);
    delete h;
    return fail(nullptr, CNS_ERR_HIP, "device/stream init: " + m);
  }
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && cus > 0) h->num_cus = (u32)cus;
    else (void)hipGetLastError();
  }
  for (auto& ev : h->ev)
    if ((e = hipEventCreate(&ev)) != hipSuccess) {
      std::string m = hipGetErrorString(e);
      cns_destroy(h);
      return fail(nullptr, CNS_ERR_HIP, "hipEventCreate: " + m);
    }
  if ((e = hipStreamCreate(&h->stream2)) != hipSuccess || (e = hipEventCreateWithFlags(&h->ev2[0], hipEventDisableTiming)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&h->ev2[1], hipEventDisableTiming)) != hipSuccess) {
    std::string m = hipGetErrorString(e);
    cns_destroy(h);
    return fail(nullptr, CNS_ERR_HIP, "second stream: " + m);
  }
  *out = h;
  return CNS_OK;
}

void cns_destroy(cns_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  for (DevBuf* b : {&h->d_part_off, &h->d_slot_node, &h->d_type_total,
                    &h->d_blocks, &h->d_cost, &h->d_fcpu, &h->d_fmem, &h->d_fcnt, &h->d_dipt, &h->d_dipcm, &h->d_dipg, &h->d_rn_off,
                    &h->d_rn_end, &h->d_rn_res, &h->d_heap, &h->d_bfj, &h->d_gupd, &h->d_fault, &h->d_pj_off, &h->d_jobs,
                    &h->d_incl, &h->d_excl, &h->d_reason_init, &h->d_results, &h->d_params, &h->d_prof, &h->d_wide, &h->d_slot_total,
                    &h->d_slot_end, &h->d_slot_type, &h->d_rv_off, &h->d_rv_start, &h->d_rv_end, &h->d_rv_res,
                    &h->d_first_resv, &h->d_resv_se})
    b->release();
  for (DevBuf* b : {&h->d_slot_block, &h->d_sib_off, &h->d_sib, &h->d_type_tag, &h->d_jtag, &h->d_params2, &h->d_pmap_a, &h->d_pmap_b, &h->d_wide_last, &h->d_params3, &h->d_pmap_c, &h->d_wide_mem, &h->d_flen, &h->d_tag_off, &h->d_tag_base}) b->release();
  for (PinBuf* b : {&h->h_place, &h->h_grouped, &h->h_reason, &h->h_jtag}) b->release();
  for (void* p : h->host_bufs) (void)hipHostFree(p);   // cns_host_alloc
  h->host_bufs.clear();
  for (DevBuf& b : h->d_prio) b.release();
  for (DevBuf& b : h->d_lim) b.release();
  for (DevBuf& b : h->d_raw) b.release();
  for (DevBuf& b : h->d_limpar) b.release();
  for (DevBuf& b : h->d_step) b.release();
  for (DevBuf& b : h->d_pre) b.release();
  h->d_gather.release();
  if (h->comm) (void)ncclCommDestroy((ncclComm_t)h->comm);
  for (auto& ev : h->ev) if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : h->ev2) if (ev) (void)hipEventDestroy(ev);
  if (h->stream2) (void)hipStreamDestroy(h->stream2);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int cns_set_nodes(cns_handle* h, const cns_node_soa* nd) {
  if (!h || !nd) return fail(h, CNS_ERR_INVALID_ARG, "cns_set_nodes: null argument");
  if (!nd->cpu_total_raw || !nd->mem_total || !nd->core_lo || !nd->part_offsets || (!nd->part_nodes && nd->part_offsets[nd->num_partitions]))
    return fail(h, CNS_ERR_INVALID_ARG, "cns_set_nodes: missing array");
  if (nd->num_nodes == 0 || nd->num_partitions == 0) return fail(h, CNS_ERR_INVALID_ARG, "cns_set_nodes: empty cluster");
  HIPCHK(h, hipSetDevice(h->device));
  if (int rc = build_gres(h, nd->gres)) return rc;
  h->have_nodes = h->have_jobs = h->have_run = false;
  const u32 N = nd->num_nodes;
  u32 P = nd->num_partitions;
  std::vector<Res> total(N);
  u64 all_gres = 0;
  for (u32 c = 0; c < h->gres.num_classes; ++c) all_gres |= h->gres.class_mask[c];
  bool big = false, wide = false;
  for (u32 n = 0; n < N; ++n) {
    total[n].cpu = nd->cpu_total_raw[n];
    total[n].mem = nd->mem_total[n];
    total[n].clo = nd->core_lo[n];
    total[n].chi = nd->core_hi ? nd->core_hi[n] : 0;
    total[n].c2 = nd->core_w2 ? nd->core_w2[n] : 0;
    total[n].c3 = nd->core_w3 ? nd->core_w3[n] : 0;
    if (total[n].c2 | total[n].c3) wide = true;
    total[n].gres = nd->gres_slots ? nd->gres_slots[n] : 0;
    if (total[n].gres & ~all_gres) return fail(h, CNS_ERR_INVALID_ARG, "node GRES slot outside every class");
    if (total[n].gres || total[n].chi) big = true;
  }
  // partitions: schedulable nodes only, ascending dense index (= canonical cost tie-break).  Partitions that share a
  // node are merged into one engine partition (union-find over the shared nodes); without sharing the engine
  // partitions are the caller's, one to one.
  const u32 total_pos = nd->part_offsets[P];
  std::vector<std::vector<std::pair<u32, u32>>> plist(P);  // per caller partition: (node, original position)
  std::vector<u32> uf(P);
  for (u32 p = 0; p < P; ++p) uf[p] = p;
  auto find = [&](u32 x) { while (uf[x] != x) { uf[x] = uf[uf[x]]; x = uf[x]; } return x; };
  std::vector<u32> first_part(N, kNone);
  bool shared = false;
  // What lies outside the engine's limits refuses ONLY the partitions it touches — the group of partitions connected through shared
  // nodes that lists the node (the reference bounds none of this: CpuSet is a std::set<uint32_t>, GRES maps are unbounded,
  // PublicHeader.h:555-573,427-494): a node the caller flags as not expressible in this ABI's formats (cns_node_soa::unsupported:
  // a core id >= 256, more GRES slots than the 64-bit mask holds), a node whose cpu count does not fit, the 65th distinct res_total
  // record, a group wider than the widest tile.  Their jobs come back with CNS_REASON_ENGINE_REFUSED; the caller's CPU scheduler takes them.
  std::vector<uint8_t> part_bad(P, 0);
  for (u32 p = 0; p < P; ++p) {
    if (nd->part_offsets[p + 1] < nd->part_offsets[p]) return fail(h, CNS_ERR_INVALID_ARG, "part_offsets not monotone");
    auto& lst = plist[p];
    for (u32 i = nd->part_offsets[p]; i < nd->part_offsets[p + 1]; ++i) {
      u32 n = nd->part_nodes[i];
      if (n >= N) return fail(h, CNS_ERR_INVALID_ARG, "part_nodes entry >= num_nodes");
      if (nd->schedulable && !nd->schedulable[n]) continue;  // JobScheduler.cpp:6595
      if ((nd->unsupported && nd->unsupported[n]) || total[n].cpu <= 0 || total[n].cpu >= 0x7FFFFFFEll) {
        part_bad[p] = (nd->unsupported && nd->unsupported[n]) ? CNS_PART_REFUSED_NODE : CNS_PART_REFUSED_CPU;
        if (first_part[n] == kNone) first_part[n] = p;   // (the partitions that share this node go with it)
        else { u32 a = find(first_part[n]), b = find(p); if (a != b) uf[std::max(a, b)] = std::min(a, b); }
        continue;
      }
      lst.emplace_back(n, i);
    }
    std::sort(lst.begin(), lst.end());
    for (size_t i = 1; i < lst.size(); ++i)
      if (lst[i].first == lst[i - 1].first) return fail(h, CNS_ERR_INVALID_ARG, "node listed twice in one partition");
    for (auto& [n, pos] : lst) {
      if (first_part[n] == kNone) first_part[n] = p;
      else { shared = true; u32 a = find(first_part[n]), b = find(p); if (a != b) uf[std::max(a, b)] = std::min(a, b); }
    }
  }
  std::vector<u32> upart_eng(P), upart_size(P);
  std::vector<uint8_t> upart_tag(P, 0);
  std::vector<std::vector<u32>> members;  // engine partition -> caller partitions, ascending
  {
    std::vector<u32> eng_of_root(P, kNone);
    for (u32 p = 0; p < P; ++p) {
      const u32 r = find(p);
      if (eng_of_root[r] == kNone) { eng_of_root[r] = (u32)members.size(); members.emplace_back(); }
      upart_eng[p] = eng_of_root[r];
      if (members[upart_eng[p]].size() >= 255) return fail(h, CNS_ERR_UNSUPPORTED, "more than 255 partitions connected through shared nodes");
      upart_tag[p] = (uint8_t)members[upart_eng[p]].size();
      members[upart_eng[p]].push_back(p);
      upart_size[p] = (u32)plist[p].size();
    }
  }
  const u32 PE = (u32)members.size();
  // ---- refusals, group by group (in engine-partition order: which group gets the last free node type is deterministic) ----
  std::vector<uint8_t> upart_refused(P, 0);
  {
    std::set<std::tuple<i64, u64, u64, u64, u64, u64, u64>> types;
    for (u32 e = 0; e < PE; ++e) {
      uint8_t why = 0;
      for (u32 p : members[e]) if (part_bad[p] && !why) why = part_bad[p];
      u32 npe = 0;
      for (u32 p : members[e]) npe += (u32)plist[p].size();
      const u32 cap = members[e].size() > 1 ? w8::WideInfo::mem_slots : std::max<u32>(kScan * (u32)CNS_NPL_MAX, w64::WideInfo::lanes * w64::WideInfo::npl_max);
      if (!why && npe > cap) why = CNS_PART_REFUSED_WIDTH;
      if (!why) {
        auto mine = types;
        for (u32 p : members[e])
          for (auto& [n, pos] : plist[p]) mine.insert(std::make_tuple(total[n].cpu, total[n].mem, total[n].clo, total[n].chi, total[n].gres, total[n].c2, total[n].c3));
        if (mine.size() > CNS_MAX_NODE_TYPES) why = CNS_PART_REFUSED_TYPES;
        else types.swap(mine);
      }
      if (why)
        for (u32 p : members[e]) { upart_refused[p] = why; plist[p].clear(); upart_size[p] = 0; }
    }
    bool any_served = false;
    for (u32 p = 0; p < P; ++p) any_served = any_served || !upart_refused[p];
    h->refused_probe = upart_refused;   // (why, per partition of THIS call: cns_group_set_nodes reads it when a device's whole share is refused)
    if (!any_served) return fail(h, CNS_ERR_UNSUPPORTED, "every partition of the snapshot is outside the engine's limits (a node flagged unsupported, a cpu count outside (0, 2^31-2), more than 64 distinct res_total records, or a group wider than the widest tile)");
  }
  std::vector<u32> part_off(PE + 1, 0), slot_node, node_slot(N, kNone);
  std::vector<std::vector<u32>> node_slots(N);
  std::vector<uint8_t> slot_tag;
  std::vector<u32> orig_pos_slot(total_pos, kNone);
  u32 max_np = 0;
  for (u32 e = 0; e < PE; ++e) {
    part_off[e] = (u32)slot_node.size();
    for (u32 p : members[e])
      for (auto& [n, pos] : plist[p]) {
        const u32 q = (u32)slot_node.size();
        if (node_slot[n] == kNone) node_slot[n] = q;
        node_slots[n].push_back(q);
        orig_pos_slot[pos] = q;
        slot_node.push_back(n);
        slot_tag.push_back(upart_tag[p]);
      }
    max_np = std::max<u32>(max_np, (u32)slot_node.size() - part_off[e]);
  }
  part_off[PE] = (u32)slot_node.size();
  // the slots of a group are its member partitions' lists one after the other: member t (its tag) owns [tag_off[b + t], tag_off[b + t + 1])
  // relative to the group's first slot, b = tag_base[group]
  h->tag_off.clear(); h->tag_base.assign(PE, 0);
  for (u32 e = 0; e < PE; ++e) {
    h->tag_base[e] = (u32)h->tag_off.size();
    u32 o = 0;
    for (u32 p : members[e]) { h->tag_off.push_back(o); o += (u32)plist[p].size(); }
    h->tag_off.push_back(o);
  }
  const u32 S = (u32)slot_node.size();
  // k_select / k_pipe tiles hold 16 576 / 8 192 slots; k_wide (64 scanner waves x 16 rows) 65 536 — but only partitions that
  // share no node with another one run on it (groups run on k_select), and only while the device can hold its workgroups
  for (u32 e = 0; e < PE; ++e) {
    const u32 npe = part_off[e + 1] - part_off[e];
    // (groups wider than k_select's tile run on k_wide's home workgroup alone, KParams::serial_only: launch_mem)
    const u32 cap = members[e].size() > 1 ? w8::WideInfo::mem_slots : std::max<u32>(kScan * (u32)CNS_NPL_MAX, w64::WideInfo::lanes * w64::WideInfo::npl_max);
    if (npe > cap)
      return fail(h, CNS_ERR_UNSUPPORTED, (members[e].size() > 1 ? "group of partitions sharing nodes with more than " : "partition with more than ") +
                                              std::to_string(cap) + " schedulable (partition, node) slots");
  }
  h->Pu = P; h->shared = shared; h->upart_eng = upart_eng; h->upart_size = upart_size; h->upart_tag = upart_tag;
  h->upart_refused = upart_refused;
  h->eng_members.clear();
  for (const auto& m : members) h->eng_members.push_back((u32)m.size());
  h->node_slots = node_slots; h->slot_tag = slot_tag;
  P = PE;
  h->N = N; h->P = P; h->S = S; h->max_np = max_np; h->big_nodes = big || wide; h->wide_cores = wide;
  h->P_real = P; h->S_real = S; h->V = 0;
  h->part_off = part_off; h->slot_node = slot_node; h->node_slot = node_slot; h->orig_pos_slot = orig_pos_slot;
  h->node_total = total;
  h->resv_start.clear(); h->resv_end.clear(); h->resv_node_slot.clear();
  h->rv_off.assign(S + 1, 0); h->rv_start.clear(); h->rv_endt.clear(); h->rv_res.clear();
  if (int rc = finalize_layout(h)) return rc;
  h->have_nodes = true;
  return CNS_OK;
}

int cns_set_reservations(cns_handle* h, const cns_resv_soa* rv) {
  if (!h) return fail(h, CNS_ERR_INVALID_ARG, "cns_set_reservations: null handle");
  if (!h->have_nodes) return fail(h, CNS_ERR_STATE, "cns_set_reservations before cns_set_nodes");
  HIPCHK(h, hipSetDevice(h->device));
  h->have_jobs = h->have_run = false;
  const u32 V = rv ? rv->num_resv : 0;
  if (V && (!rv->start_sec || !rv->end_sec || !rv->alloc_offsets || !rv->alloc_node || !rv->alloc_cpu_raw ||
            !rv->alloc_mem || !rv->alloc_core_lo))
    return fail(h, CNS_ERR_INVALID_ARG, "cns_set_reservations: missing array");
  // back to the layout of cns_set_nodes, then append one virtual partition per reservation
  h->P = h->P_real; h->S = h->S_real; h->V = V;
  h->part_off.resize(h->P_real + 1);
  h->slot_node.resize(h->S_real);
  h->resv_start.assign(V, 0); h->resv_end.assign(V, 0);
  h->resv_node_slot.assign(V, {});
  std::vector<std::vector<std::tuple<i64, i64, Res>>> per_slot(h->S_real);  // reservation entries of the real slots
  std::vector<Res> virt_total;
  u64 all_gres = 0;
  for (u32 c = 0; c < h->gres.num_classes; ++c) all_gres |= h->gres.class_mask[c];
  for (u32 v = 0; v < V; ++v) {
    h->resv_start[v] = rv->start_sec[v];
    h->resv_end[v] = rv->end_sec[v];
    if (rv->alloc_offsets[v + 1] < rv->alloc_offsets[v]) return fail(h, CNS_ERR_INVALID_ARG, "reservation alloc_offsets not monotone");
    std::vector<std::pair<u32, Res>> al;
    for (u32 a = rv->alloc_offsets[v]; a < rv->alloc_offsets[v + 1]; ++a) {
      const u32 n = rv->alloc_node[a];
      if (n >= h->N) return fail(h, CNS_ERR_INVALID_ARG, "reservation node >= num_nodes");
      Res r;
      r.cpu = rv->alloc_cpu_raw[a]; r.mem = rv->alloc_mem[a]; r.clo = rv->alloc_core_lo[a];
      r.chi = rv->alloc_core_hi ? rv->alloc_core_hi[a] : 0;
      r.c2 = rv->alloc_core_w2 ? rv->alloc_core_w2[a] : 0;
      r.c3 = rv->alloc_core_w3 ? rv->alloc_core_w3[a] : 0;
      r.gres = rv->alloc_gres ? rv->alloc_gres[a] : 0;
      if (r.gres & ~all_gres) return fail(h, CNS_ERR_INVALID_ARG, "reservation GRES slot outside every class");
      if (r.cpu <= 0 || r.cpu >= 0x7FFFFFFEll) return fail(h, CNS_ERR_UNSUPPORTED, "reservation cpu share must be in (0, 2^31-2)");
      al.emplace_back(n, r);
      if (r.gres || r.chi) h->big_nodes = true;
    }
    std::sort(al.begin(), al.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    for (size_t i = 1; i < al.size(); ++i)
      if (al[i].first == al[i - 1].first) return fail(h, CNS_ERR_INVALID_ARG, "node listed twice in one reservation");
    for (auto& [n, r] : al) {
      h->resv_node_slot[v][n] = (u32)h->slot_node.size();  // virtual node: its own NodeState (:6661-6664)
      h->slot_node.push_back(n);
      virt_total.push_back(r);
      for (u32 q : h->node_slots[n]) per_slot[q].emplace_back(rv->start_sec[v], rv->end_sec[v], r);   // every partition's slot of the node
    }
    h->part_off.push_back((u32)h->slot_node.size());
    h->max_np = std::max<u32>(h->max_np, (u32)al.size());
  }
  h->P = h->P_real + V;
  h->S = (u32)h->slot_node.size();
  h->rv_off.assign(h->S + 1, 0); h->rv_start.clear(); h->rv_endt.clear(); h->rv_res.clear();
  for (u32 q = 0; q < h->S_real; ++q) {
    // release + dip events of a node share the node block's upper half with the running allocations (k_init_nodes)
    if (per_slot[q].size() > 200) return fail(h, CNS_ERR_UNSUPPORTED, "more than 200 reservations on one node");
    for (auto& [st, en, r] : per_slot[q]) { h->rv_start.push_back(st); h->rv_endt.push_back(en); h->rv_res.push_back(r); }
    h->rv_off[q + 1] = (u32)h->rv_start.size();
  }
  for (u32 q = h->S_real; q < h->S; ++q) h->rv_off[q + 1] = h->rv_off[q];
  if (int rc = finalize_layout(h, &virt_total)) return rc;
  // running jobs must be set again after the layout changed
  std::vector<u32> rn_off(h->S + 1, 0);
  if (int rc = upload(h, h->d_rn_off, rn_off)) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return CNS_OK;
}

int cns_set_running(cns_handle* h, const cns_running_soa* rn) {
  if (!h) return fail(h, CNS_ERR_INVALID_ARG, "cns_set_running: null handle");
  if (!h->have_nodes) return fail(h, CNS_ERR_STATE, "cns_set_running before cns_set_nodes");
  HIPCHK(h, hipSetDevice(h->device));
  const u32 N = h->N, S = h->S;
  // allocations are grouped by SLOT: the node's own slot, or — for a job running inside a reservation
  // (JobScheduler.cpp:6692-6707) — the reservation's virtual node
  // (slots of the node: one per partition that lists it — they all start from the same NodeState, JobScheduler.h:498-511)
  static const std::vector<u32> kNoSlots;
  std::vector<u32> one(1);
  auto slots_of = [&](u32 job, u32 n) -> const std::vector<u32>& {
    const u32 v = rn->reservation ? rn->reservation[job] : CNS_RESV_NONE;
    if (v == CNS_RESV_NONE) return h->node_slots[n];  // empty: unschedulable node, ignored (:6685-6686)
    if (v >= h->V) return kNoSlots;                    // reservation not found (:6693-6700)
    auto it = h->resv_node_slot[v].find(n);
    if (it == h->resv_node_slot[v].end()) return kNoSlots;
    one[0] = it->second;
    return one;
  };
  std::vector<u32> rn_off(S + 1, 0);
  std::vector<i64> rn_end;
  std::vector<Res> rn_res;
  if (rn && rn->num_jobs) {
    if (!rn->end_sec || !rn->alloc_offsets || !rn->alloc_node || !rn->alloc_cpu_raw || !rn->alloc_mem || !rn->alloc_core_lo)
      return fail(h, CNS_ERR_INVALID_ARG, "cns_set_running: missing array");
    const u32 A = rn->alloc_offsets[rn->num_jobs];
    if (A != rn->num_allocs) return fail(h, CNS_ERR_INVALID_ARG, "cns_set_running: num_allocs mismatch");
    for (u32 j = 0; j < rn->num_jobs; ++j)
      for (u32 a = rn->alloc_offsets[j]; a < rn->alloc_offsets[j + 1]; ++a) {
        const u32 n = rn->alloc_node[a];
        if (n >= N) return fail(h, CNS_ERR_INVALID_ARG, "running allocation on node >= num_nodes");
        for (u32 q : slots_of(j, n)) rn_off[q + 1]++;
      }
    for (u32 q = 0; q < S; ++q) {
      const u32 nrv = h->rv_off[q + 1] - h->rv_off[q];
      if (nrv ? rn_off[q + 1] + 2 * nrv + 2 > kTlCap / 2 : rn_off[q + 1] + 2 > kTlCap)
        return fail(h, CNS_ERR_UNSUPPORTED, "too many running allocations / reservations on one node (1006, or 502 events with reservations)");
      rn_off[q + 1] += rn_off[q];
    }
    rn_end.resize(rn_off[S]);
    rn_res.resize(rn_off[S]);
    h->ent_job.assign(rn_off[S], 0);
    h->ent_slot.assign(rn_off[S], 0);
    std::vector<u32> cur(rn_off.begin(), rn_off.end() - 1);
    for (u32 j = 0; j < rn->num_jobs; ++j)  // stable: per slot, input order (cost accumulation order)
      for (u32 a = rn->alloc_offsets[j]; a < rn->alloc_offsets[j + 1]; ++a) {
        Res r;
        r.cpu = rn->alloc_cpu_raw[a];
        r.mem = rn->alloc_mem[a];
        r.clo = rn->alloc_core_lo[a];
        r.chi = rn->alloc_core_hi ? rn->alloc_core_hi[a] : 0;
        r.c2 = rn->alloc_core_w2 ? rn->alloc_core_w2[a] : 0;
        r.c3 = rn->alloc_core_w3 ? rn->alloc_core_w3[a] : 0;
        r.gres = rn->alloc_gres ? rn->alloc_gres[a] : 0;
        for (u32 q : slots_of(j, rn->alloc_node[a])) {
          u32 d = cur[q]++;
          rn_end[d] = rn->end_sec[j];
          rn_res[d] = r;
          h->ent_job[d] = j;
          h->ent_slot[d] = q;
        }
      }
  }
  if (!(rn && rn->num_jobs)) { h->ent_job.clear(); h->ent_slot.clear(); }
  h->R = rn ? rn->num_jobs : 0;
  h->ent_end = rn_end;
  if (int rc = upload(h, h->d_rn_off, rn_off)) return rc;
  if (rn_end.empty()) { rn_end.push_back(0); rn_res.push_back(Res{0, 0, 0, 0, 0}); }
  if (int rc = upload(h, h->d_rn_end, rn_end)) return rc;
  if (int rc = upload(h, h->d_rn_res, rn_res)) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->have_run = false;
  return CNS_OK;
}

static int upload_jobs_impl(cns_handle* h, const cns_job_soa* jb);
// The caller owns its arrays again when the call is back — on EVERY path: an error behind the first asynchronous copy (a failed allocation, a
// queue that fails validation) returns only after the copies from the caller's arrays have drained.
int cns_upload_jobs(cns_handle* h, const cns_job_soa* jb) {
  const int rc = upload_jobs_impl(h, jb);
  if (rc != 0 && h && h->have_nodes) {
    const std::string keep = h->err;            // (the drain must not replace the error it follows)
    if (hipSetDevice(h->device) == hipSuccess) (void)hipStreamSynchronize(h->stream);
    (void)hipGetLastError();
    h->err = keep;
  }
  return rc;
}
static int upload_jobs_impl(cns_handle* h, const cns_job_soa* jb) {
  if (!h || !jb) return fail(h, CNS_ERR_INVALID_ARG, "cns_upload_jobs: null argument");
  if (!h->have_nodes) return fail(h, CNS_ERR_STATE, "cns_upload_jobs before cns_set_nodes");
  const u64 J = jb->num_jobs;
  if (J && (!jb->partition || !jb->time_limit_sec || !jb->node_mem || !jb->task_cpu_raw || !jb->task_mem ||
            !jb->node_num || !jb->ntasks || !jb->ntasks_per_node_min || !jb->ntasks_per_node_max))
    return fail(h, CNS_ERR_INVALID_ARG, "cns_upload_jobs: missing array");
  if (J > 0xFFFFFFF0ull) return fail(h, CNS_ERR_UNSUPPORTED, "more than 2^32-16 jobs");
  HIPCHK(h, hipSetDevice(h->device));
  hipEvent_t e0 = h->ev[0], e1 = h->ev[1];
  HIPCHK(h, hipEventRecord(e0, h->stream));
  h->have_jobs = h->have_run = false;
  const u64 batch = h->cfg.scheduled_batch_size ? std::min<u64>(h->cfg.scheduled_batch_size, J) : J;
  // The caller's arrays go to the device as they are (k_pack_jobs builds the 32-dword job records there); from page-locked arrays
  // (cns_host_alloc) these are DMA transfers the thread does not wait for, and both host passes below run in their shadow.  A queue
  // that fails validation returns only after the transfers have drained: the caller owns its arrays again when the call is back.
  // (offsets without their node list are reported below, after the checks that come first: raw() issues no copy from a null source)
  auto raw = [&](DevBuf& d, const void* src, size_t bytes) -> int {
    HIPCHK(h, d.ensure(bytes));
    if (src && bytes) HIPCHK(h, hipMemcpyAsync(d.p, src, bytes, hipMemcpyHostToDevice, h->stream));
    return 0;
  };
  DevBuf* rb_ = h->d_raw;  // 0 L, 1 ncpu, 2 nmem, 3 tcpu, 4 tmem, 5 k, 6 ntasks, 7 tmin, 8 tmax, 9 excl, 10 gtot, 11 gspec, 12 incl_off,
                           // 13 excl_off, 14 place_off, 15 grouped
  if (int rc = raw(rb_[0], jb->time_limit_sec, J * 8)) return rc;
  if (jb->node_cpu_raw) { if (int rc = raw(rb_[1], jb->node_cpu_raw, J * 8)) return rc; }
  if (int rc = raw(rb_[2], jb->node_mem, J * 8)) return rc;
  if (int rc = raw(rb_[3], jb->task_cpu_raw, J * 8)) return rc;
  if (int rc = raw(rb_[4], jb->task_mem, J * 8)) return rc;
  if (int rc = raw(rb_[5], jb->node_num, J * 4)) return rc;
  if (int rc = raw(rb_[6], jb->ntasks, J * 4)) return rc;
  if (int rc = raw(rb_[7], jb->ntasks_per_node_min, J * 4)) return rc;
  if (int rc = raw(rb_[8], jb->ntasks_per_node_max, J * 4)) return rc;
  if (jb->exclusive) { if (int rc = raw(rb_[9], jb->exclusive, J)) return rc; }
  if (jb->gres_total) { if (int rc = raw(rb_[10], jb->gres_total, J * CNS_MAX_GRES_NAMES)) return rc; }
  if (jb->gres_spec) { if (int rc = raw(rb_[11], jb->gres_spec, J * CNS_MAX_GRES_CLASSES)) return rc; }
  if (jb->incl_offsets) { if (int rc = raw(rb_[12], jb->incl_offsets, (J + 1) * 8)) return rc; }
  if (jb->excl_offsets) { if (int rc = raw(rb_[13], jb->excl_offsets, (J + 1) * 8)) return rc; }
  const u64 n_incl = jb->incl_offsets ? jb->incl_offsets[J] : 0, n_excl = jb->excl_offsets ? jb->excl_offsets[J] : 0;
  HIPCHK(h, h->d_incl.ensure(std::max<u64>(n_incl, 1) * 4));
  HIPCHK(h, h->d_excl.ensure(std::max<u64>(n_excl, 1) * 4));
  if (int rc = raw(h->d_incl, jb->incl_nodes, n_incl * 4)) return rc;
  if (int rc = raw(h->d_excl, jb->excl_nodes, n_excl * 4)) return rc;
  // BasicPriority (JobScheduler.h:185-200) + per-job pre-checks of the ordered loop (cpp:6744-6761): jobs_host.inc, pass 1 — on a few host threads
  namespace jh = cns_jobs_host;
  HIPCHK(h, h->h_reason.ensure(std::max<u64>(J, 1)));
  HIPCHK(h, h->h_place.ensure((J + 1) * 8));
  if (h->shared) HIPCHK(h, h->h_jtag.ensure(std::max<u64>(J, 1)));
  h->job_part.resize((size_t)J);
  jh::Route R;
  R.P = h->P; R.Pu = h->Pu; R.P_real = h->P_real; R.V = h->V;
  R.upart_refused = h->upart_refused.data(); R.upart_eng = h->upart_eng.data(); R.upart_size = h->upart_size.data();
  R.upart_tag = h->upart_tag.data(); R.part_off = h->part_off.data();
  R.s_node = h->big_nodes ? 48 : 32; R.gres_classes = h->gres.num_classes; R.batch = batch;
  jh::Out O;
  O.reason = h->h_reason.as<uint8_t>(); O.job_part = h->job_part.data(); O.place_off = h->h_place.as<u64>();
  O.jtag = h->shared ? h->h_jtag.as<uint8_t>() : nullptr;
  h->place_off = O.place_off;
  std::vector<jh::Chunk> chunks;
  {
    std::string perr;
    if (const int rc = jh::pass1(jb, R, O, chunks, jh::threads_for(J, h->host_threads), &perr)) { (void)hipStreamSynchronize(h->stream); return fail(h, rc, perr); }
  }
  if ((jb->incl_offsets && !jb->incl_nodes && jb->incl_offsets[J]) || (jb->excl_offsets && !jb->excl_nodes && jb->excl_offsets[J])) {
    (void)hipStreamSynchronize(h->stream);
    return fail(h, CNS_ERR_INVALID_ARG, "cns_upload_jobs: include / exclude offsets without node lists");
  }
  const u64 Jg = O.Jg, places = O.places;
  const std::vector<u64>& pj_off = O.pj_off;
  h->part_jobs.resize(h->P);
  for (u32 p = 0; p < h->P; ++p) h->part_jobs[p] = pj_off[p + 1] - pj_off[p];
  HIPCHK(h, h->h_grouped.ensure(std::max<u64>(Jg, 1) * 4));
  O.grouped = h->h_grouped.as<u32>();
  // pass 2: the offsets of the placement records and the queue grouped by partition in queue order, one u32 per job
  jh::pass2(jb, O, chunks);
  if (J == 0) { O.reason[0] = CNS_REASON_NONE; if (O.jtag) O.jtag[0] = 0; }   // (the one-element stand-ins of an empty queue)
  if (Jg == 0) O.grouped[0] = 0;
  if (int rc = raw(rb_[14], h->h_place.p, (J + 1) * 8)) return rc;
  if (int rc = raw(rb_[15], h->h_grouped.p, std::max<u64>(Jg, 1) * 4)) return rc;
  if (h->shared) { if (int rc = raw(h->d_jtag, h->h_jtag.p, std::max<u64>(J, 1))) return rc; }
  HIPCHK(h, h->d_jobs.ensure((size_t)std::max<u64>(Jg, 1) * kJobRecDwords * 4));
  if (Jg) {
    PackParams K{};
    K.Jg = Jg; K.grouped = rb_[15].as<u32>();
    K.L = rb_[0].as<i64>(); K.ncpu = jb->node_cpu_raw ? rb_[1].as<i64>() : nullptr; K.nmem = rb_[2].as<u64>();
    K.tcpu = rb_[3].as<i64>(); K.tmem = rb_[4].as<u64>(); K.k = rb_[5].as<u32>(); K.ntasks = rb_[6].as<u32>();
    K.tmin = rb_[7].as<u32>(); K.tmax = rb_[8].as<u32>();
    K.excl = jb->exclusive ? rb_[9].as<uint8_t>() : nullptr;
    K.gtot = jb->gres_total ? rb_[10].as<uint8_t>() : nullptr; K.gspec = jb->gres_spec ? rb_[11].as<uint8_t>() : nullptr;
    K.incl_off = jb->incl_offsets ? rb_[12].as<u64>() : nullptr; K.excl_off = jb->excl_offsets ? rb_[13].as<u64>() : nullptr;
    K.place_off = rb_[14].as<u64>(); K.jobrec = h->d_jobs.as<u32>();
    K.tag = h->shared ? h->d_jtag.as<uint8_t>() : nullptr;
    hipLaunchKernelGGL(k_pack_jobs, dim3((unsigned)((Jg + 255) / 256)), dim3(256), 0, h->stream, K);
    HIPCHK(h, hipGetLastError());
  }

  if (int rc = upload(h, h->d_pj_off, pj_off)) return rc;
  if (int rc = raw(h->d_reason_init, h->h_reason.p, std::max<u64>(J, 1))) return rc;
  // results: one contiguous HBM buffer (also what an RCCL allgather ships)
  cns_engine::ResOff& r = h->ro;
  size_t ro = 0;
  auto rsec = [&](size_t elem, u64 n) { size_t x = ro; ro = align16(ro + elem * (size_t)std::max<u64>(n, 1)); return x; };
  r.start = rsec(8, J); r.cpu = rsec(8, places); r.mem = rsec(8, places); r.clo = rsec(8, places);
  r.chi = rsec(8, places); r.gres = rsec(8, places); r.node = rsec(4, places); r.ntasks = rsec(4, places);
  r.reason = rsec(1, J);
  r.c2 = r.c3 = ro;   // the planes of core ids 128..255 exist only for a snapshot with such nodes (nothing more to ship otherwise)
  if (h->wide_cores) { r.c2 = rsec(8, places); r.c3 = rsec(8, places); }
  r.total = ro;
  HIPCHK(h, h->d_results.ensure(ro));
  HIPCHK(h, hipEventRecord(e1, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  float ms = 0;
  HIPCHK(h, hipEventElapsedTime(&ms, e0, e1));
  h->timing = cns_timing{};
  h->timing.h2d_ms = ms;
  h->J = J; h->Jg = Jg; h->places = places; h->jobs_ordered = batch; h->algo_bytes = O.algo; h->window_shaped = O.n_shaped;
  h->have_jobs = true;
  return CNS_OK;
}

// device buffers of a cycle with preemption (cns_engine::d_pre)
enum { B_QPOFF, B_QP, B_PJQOS, B_PJQP, B_PJPRIO, B_PJREC0, B_PJK, B_PJEND, B_RNJOB, B_ENTSLOT, B_ENTGONE, B_RJQOS, B_RJQP,
       B_RJSTART, B_RJEND, B_RJPRE, B_RJOFF, B_RJENT, B_HEAD, B_RECNEXT, B_RECORIG, B_RECSLOT, B_RECGONE, B_MISC };

// One pass of the cycle on the device.  *fault_code: the device fault it ended with (0: none).
static int run_resident_once(cns_handle* h, int64_t now, u32* fault_code) {
  *fault_code = 0;
  HIPCHK(h, hipSetDevice(h->device));
  KParams K;
  fill_params(h, K, now);
  char* rb = h->d_results.as<char>();
  const u64 pl = std::max<u64>(h->places, 1), J = std::max<u64>(h->J, 1);
  HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
  HIPCHK(h, hipMemsetAsync(rb + h->ro.start, 0, h->ro.node - h->ro.start, h->stream));  // start + 8-byte records
  HIPCHK(h, hipMemsetAsync(rb + h->ro.node, 0xFF, 4 * pl, h->stream));                  // CNS_NODE_NONE
  HIPCHK(h, hipMemsetAsync(rb + h->ro.ntasks, 0, 4 * pl, h->stream));
  if (h->wide_cores) HIPCHK(h, hipMemsetAsync(rb + h->ro.c2, 0, h->ro.total - h->ro.c2, h->stream));   // core ids 128..255 of the records
  HIPCHK(h, hipMemcpyAsync(rb + h->ro.reason, h->d_reason_init.p, J, hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(h, hipMemsetAsync(h->d_fault.p, 0, 16, h->stream));
  if (h->pre_active) {
    // The mutable preemption state starts every PASS empty, not every call: the retry after a k_wide protocol fault re-runs the
    // k_select partitions too, and a second pass over a first pass's per-slot job lists (slot_head / rec_next), hidden
    // candidates (ent_gone / rec_gone) and preempted pairs (out_cnt) would loop on a self-linked list or report pairs twice.
    DevBuf* B = h->d_pre;
    HIPCHK(h, hipMemsetAsync(B[B_ENTGONE].p, 0, std::max<size_t>(h->ent_job.size(), 1), h->stream));
    HIPCHK(h, hipMemsetAsync(B[B_HEAD].p, 0xFF, (size_t)std::max<u32>(h->S, 1) * 4, h->stream));
    HIPCHK(h, hipMemsetAsync(B[B_RECGONE].p, 0, pl, h->stream));
    HIPCHK(h, hipMemsetAsync(h->pre_params.out_cnt, 0, 16, h->stream));
  }
  HIPCHK(h, hipMemsetAsync(h->d_prof.p, 0, ((size_t)h->P * (32 + 16) + 2048) * sizeof(u64), h->stream));   // cycle counters + the always-on protocol counters
  HIPCHK(h, h->d_params.ensure(sizeof(KParams)));
  HIPCHK(h, hipMemcpyAsync(h->d_params.p, &K, sizeof(KParams), hipMemcpyHostToDevice, h->stream));
  if (h->S) hipLaunchKernelGGL(k_init_nodes, dim3((h->S + 255) / 256), dim3(256), 0, h->stream, h->d_params.as<KParams>());
  if (h->Jg) hipLaunchKernelGGL(k_prep_jobs, dim3((unsigned)((h->Jg + 255) / 256)), dim3(256), 0, h->stream, h->d_params.as<KParams>(), (u64)h->Jg);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipEventRecord(h->ev[1], h->stream));
  bool split = false;
  if (h->Jg) {
    // Which partitions need k_select: groups of partitions that share nodes (one time map per node, a cost per partition) and,
    // in a cycle with preemption, the partitions that have a pending job whose qos may preempt anything (TryPreempt_ returns at
    // JobScheduler.cpp:6384-6385 for every other job).  Everything else runs on k_wide / k_pipe IN THE SAME CYCLE, side by
    // side on a second stream: partitions with disjoint node sets never interact (:6723-6732,6746-6761).
    // Only partitions that HAVE pending jobs get a scheduler (the reference builds NodeStates and a LocalScheduler only for
    // the partitions some pending job names, JobScheduler.cpp:6516-6530,6571-6573,6723-6732): the launch, and with it the choice
    // of the k_wide build (workgroups per partition), is sized by the busy partitions, not by the snapshot.
    // ... and a group of partitions that share nodes and is wider than k_select's register tile runs on k_wide's home workgroup
    // alone (launch_mem): the ordinary "ALL partition over the whole cluster" layout of a large site.
    std::vector<u32> pa, pb, pc;
    u32 npa = 0, npb = 0, npc = 0;
    for (u32 p = 0; p < h->P; ++p) {
      if (p >= h->part_jobs.size() || h->part_jobs[p] == 0) continue;
      const bool pre = h->pre_active && p < h->pre_part.size() && h->pre_part[p];
      const bool sel = (p < h->eng_members.size() && h->eng_members[p] > 1) || pre;
      const u32 np = h->part_off[p + 1] - h->part_off[p];
      if (sel && np > kScan * (u32)CNS_NPL_MAX) {
        if (pre) return fail(h, CNS_ERR_UNSUPPORTED, "preemption among the jobs of a partition (or group of partitions sharing nodes) with more than " +
                                                         std::to_string(kScan * (u32)CNS_NPL_MAX) + " (partition, node) slots");
        pc.push_back(p); npc = std::max(npc, np);
      } else if (sel) { pb.push_back(p); npb = std::max(npb, np); }
      else { pa.push_back(p); npa = std::max(npa, np); }
    }
    std::string err, name_a, name_b, name_c;
    if (!pc.empty()) {
      if (int rc = upload(h, h->d_pmap_c, pc)) return rc;
      HIPCHK(h, h->d_params3.ensure(sizeof(KParams)));
      KParams KC = K;
      KC.part_map = h->d_pmap_c.as<u32>(); KC.launch_parts = (u32)pc.size();
      KC.general_only = 0; KC.pre = PreParams{};
      HIPCHK(h, hipEventRecord(h->ev2[0], h->stream));                  // tables + init kernels done: the second stream may start
      HIPCHK(h, hipStreamWaitEvent(h->stream2, h->ev2[0], 0));
      LaunchCtx LC{(u32)pc.size(), npc, h->stream2, h->d_params3.as<KParams>(), 0u};
      if (launch_mem(h, KC, LC, &name_c)) return fail(h, CNS_ERR_HIP, "k_mem: control block allocation / upload / launch failed");
      name_c += " on " + std::to_string(pc.size()) + " group(s) of up to " + std::to_string(npc) + " slots";
    }
    const u32 held = (u32)pc.size();   // home workgroups of k_mem that hold a CU while the other launches run
    if (pa.empty() && pb.empty()) {
      h->last_kernel = pc.empty() ? std::string("none (no pending job reaches an ordered loop)") : name_c;   // (never empty: callers parse it)
    } else if (pb.empty() || pa.empty()) {
      // one launch: over all partitions (identity map) when every one is busy, else over the busy ones (part_map)
      const bool plain = pb.empty();
      const std::vector<u32>& pm = plain ? pa : pb;
      const bool ident = pm.size() == h->P;
      LaunchCtx L{(u32)pm.size(), plain ? npa : npb, h->stream, h->d_params.as<KParams>(), held};
      KParams K1 = K;
      if (plain) { K1.general_only = 0; K1.pre = PreParams{}; }
      if (!ident) {
        if (int rc = upload(h, h->d_pmap_a, pm)) return rc;
        K1.part_map = h->d_pmap_a.as<u32>(); K1.launch_parts = (u32)pm.size();
      }
      if (K1.general_only != K.general_only || !ident) HIPCHK(h, hipMemcpyAsync(h->d_params.p, &K1, sizeof(KParams), hipMemcpyHostToDevice, h->stream));
      if (const int rc = launch_one(h, K1, L, plain, &h->last_kernel, &err)) return fail(h, rc, err);
      if (!ident) h->last_kernel += " on " + std::to_string(pm.size()) + " busy of " + std::to_string(h->P) + " partitions";
      if (!pc.empty()) h->last_kernel += " + " + name_c;
    } else {
      if (int rc = upload(h, h->d_pmap_a, pa)) return rc;
      if (int rc = upload(h, h->d_pmap_b, pb)) return rc;
      HIPCHK(h, h->d_params2.ensure(sizeof(KParams)));
      KParams KA = K, KB = K;
      KA.part_map = h->d_pmap_a.as<u32>(); KA.launch_parts = (u32)pa.size();
      KA.general_only = 0; KA.pre = PreParams{};
      KA.slot_block = nullptr; KA.sib_off = nullptr; KA.sib = nullptr; KA.slot_tag = nullptr;   // (none of these partitions shares a node)
      KB.part_map = h->d_pmap_b.as<u32>(); KB.launch_parts = (u32)pb.size();
      HIPCHK(h, hipMemcpyAsync(h->d_params.p, &KA, sizeof(KParams), hipMemcpyHostToDevice, h->stream));
      HIPCHK(h, hipMemcpyAsync(h->d_params2.p, &KB, sizeof(KParams), hipMemcpyHostToDevice, h->stream));
      HIPCHK(h, hipEventRecord(h->ev2[0], h->stream));                  // tables + init kernels done: the second stream may start
      HIPCHK(h, hipStreamWaitEvent(h->stream2, h->ev2[0], 0));
      LaunchCtx LB{(u32)pb.size(), npb, h->stream2, h->d_params2.as<KParams>(), 0u};
      if (const int rc = launch_one(h, KB, LB, false, &name_b, &err)) return fail(h, rc, err);   // first: its few workgroups take their CUs
      LaunchCtx LA{(u32)pa.size(), npa, h->stream, h->d_params.as<KParams>(), (u32)pb.size() + held};
      if (const int rc = launch_one(h, KA, LA, true, &name_a, &err)) return fail(h, rc, err);
      HIPCHK(h, hipEventRecord(h->ev[3], h->stream));                   // the partitions on the fast kernels are done here
      split = true;
      HIPCHK(h, hipEventRecord(h->ev2[1], h->stream2));
      HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev2[1], 0));           // the cycle ends when both have
      h->last_kernel = name_a + " + " + name_b + " on " + std::to_string(pb.size()) + " of " + std::to_string(h->P) + " partitions";
      if (!pc.empty()) h->last_kernel += " + " + name_c;
    }
    if (!pc.empty() && !split) {   // the cycle ends when the launch on the second stream has
      HIPCHK(h, hipEventRecord(h->ev2[1], h->stream2));
      HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev2[1], 0));
    }
    HIPCHK(h, hipGetLastError());
  }
  HIPCHK(h, hipEventRecord(h->ev[2], h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  float a = 0, b = 0;
  HIPCHK(h, hipEventElapsedTime(&a, h->ev[0], h->ev[1]));
  HIPCHK(h, hipEventElapsedTime(&b, h->ev[1], h->ev[2]));
  h->timing.init_ms = a;
  h->timing.select_ms = b;
  if (split) {   // a split cycle: when the partitions on the fast kernels were done (the cycle itself ends with the slower launch)
    float c = 0;
    HIPCHK(h, hipEventElapsedTime(&c, h->ev[1], h->ev[3]));
    char buf[64];
    snprintf(buf, sizeof buf, " [%s partitions done after %.1f ms]", h->last_kernel.substr(0, h->last_kernel.find(' ')).c_str(), c);
    h->last_kernel += buf;
  }
  h->timing.jobs_ordered = h->jobs_ordered;
  h->timing.algorithmic_bytes = h->algo_bytes;
  u32 fault[4] = {0, 0, 0, 0};
  HIPCHK(h, hipMemcpy(fault, h->d_fault.p, 16, hipMemcpyDeviceToHost));
  h->last_now = now;
  if (fault[0] && h->pre_active && (fault[0] == 31 || fault[0] == 32)) {
    *fault_code = fault[0];
    return fail(h, CNS_ERR_UNSUPPORTED, std::string("cns_select_preempt: job ") + std::to_string(fault[1]) + (fault[0] == 31
                    ? " has more preemption candidates on its nodes than the candidate buffer holds"
                    : " needs more segment-tree nodes than the per-partition pool (65536) holds") +
                    "; keep the CPU SchedulerAlgo for this cycle (include/crane_gpu/preempt.h, limits)");
  }
  if (fault[0]) {
    *fault_code = fault[0];
    return fail(h, CNS_ERR_DEVICE_FAULT, "device invariant violated: code " + std::to_string(fault[0]) + " job " +
                                             std::to_string(fault[1]) + " aux " + std::to_string(fault[2]) + "," +
                                             std::to_string(fault[3]) + " (" + h->last_kernel + ")");
  }
  h->have_run = true;
  return CNS_OK;
}

// k_wide is a persistent kernel whose workgroups wait for each other; its waits are bounded and end in a device fault
// (codes 20..41: a wait of the exchange / command / task-ring protocol ran out, or a protocol position did not match).
// Such a fault says nothing about the INPUT: the cycle is re-run once on k_pipe / k_select, which live inside one
// workgroup per partition (every run starts from the caller's tables: k_init_nodes, k_prep_jobs and the result buffers
// are part of the pass).  Faults below 20 are data invariants of the shared routines (e.g. 3: the input class on which
// the reference itself asserts) and would recur: they fail the call.
int cns_run_resident(cns_handle* h, int64_t now) {
  if (!h) return fail(h, CNS_ERR_INVALID_ARG, "cns_run_resident: null handle");
  if (!h->have_nodes || !h->have_jobs) return fail(h, CNS_ERR_STATE, "cns_run_resident before set_nodes/upload_jobs");
  u32 code = 0;
  h->wide_off = false;
  int rc = run_resident_once(h, now, &code);
  // (CNS_WIDE_NO_RETRY=1: the fault fails the call — the GPU parity tests run that way, so that a k_wide that breaks is seen
  // and not papered over by the kernels behind it)
  const char* no_retry = getenv("CNS_WIDE_NO_RETRY");
  if (rc == CNS_ERR_DEVICE_FAULT && code >= 20 && h->last_kernel.rfind("k_wide", 0) == 0 && !(no_retry && no_retry[0] == '1')) {
    const std::string first = h->err;
    h->wide_off = true;
    ++h->wide_retries;
    rc = run_resident_once(h, now, &code);
    h->wide_off = false;
    if (rc == CNS_OK) h->last_kernel += " (retry after: " + first + ")";
    else h->err = first + "; retry on " + h->last_kernel + ": " + h->err;
  }
  return rc;
}

int cns_download(cns_handle* h, cns_placement_soa* out) {
  if (!h || !out) return fail(h, CNS_ERR_INVALID_ARG, "cns_download: null argument");
  if (!h->have_run) return fail(h, CNS_ERR_STATE, "cns_download before a successful run");
  if (out->place_capacity < h->places) return fail(h, CNS_ERR_INVALID_ARG, "cns_download: place_capacity too small");
  if (!out->start_sec || !out->reason || !out->place_offsets || !out->node_idx || !out->ntasks || !out->cpu_raw ||
      !out->mem || !out->core_lo || !out->core_hi || !out->gres)
    return fail(h, CNS_ERR_INVALID_ARG, "cns_download: missing result array");
  if (h->wide_cores && (!out->core_w2 || !out->core_w3))
    return fail(h, CNS_ERR_INVALID_ARG, "cns_download: the snapshot has nodes with core ids above 127: core_w2 / core_w3 are required");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
  const char* rb = h->d_results.as<char>();
  auto get = [&](void* dst, size_t off, size_t bytes) -> hipError_t {
    return bytes ? hipMemcpyAsync(dst, rb + off, bytes, hipMemcpyDeviceToHost, h->stream) : hipSuccess;
  };
  const size_t J = (size_t)h->J, pl = (size_t)h->places;
  HIPCHK(h, get(out->start_sec, h->ro.start, 8 * J));
  HIPCHK(h, get(out->reason, h->ro.reason, J));
  HIPCHK(h, get(out->cpu_raw, h->ro.cpu, 8 * pl));
  HIPCHK(h, get(out->mem, h->ro.mem, 8 * pl));
  HIPCHK(h, get(out->core_lo, h->ro.clo, 8 * pl));
  HIPCHK(h, get(out->core_hi, h->ro.chi, 8 * pl));
  HIPCHK(h, get(out->gres, h->ro.gres, 8 * pl));
  if (h->wide_cores) {
    HIPCHK(h, get(out->core_w2, h->ro.c2, 8 * pl));
    HIPCHK(h, get(out->core_w3, h->ro.c3, 8 * pl));
  } else {
    if (out->core_w2 && pl) memset(out->core_w2, 0, 8 * pl);
    if (out->core_w3 && pl) memset(out->core_w3, 0, 8 * pl);
  }
  HIPCHK(h, get(out->node_idx, h->ro.node, 4 * pl));
  HIPCHK(h, get(out->ntasks, h->ro.ntasks, 4 * pl));
  HIPCHK(h, hipEventRecord(h->ev[1], h->stream));
  memcpy(out->place_offsets, h->place_off, 8 * (J + 1));   // (host to host, in the shadow of the transfers)
  HIPCHK(h, hipStreamSynchronize(h->stream));
  float ms = 0;
  HIPCHK(h, hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
  h->timing.d2h_ms = ms;
  return CNS_OK;
}

int cns_host_alloc(cns_handle* h, uint64_t bytes, void** out) {
  if (!h || !out) return fail(h, CNS_ERR_INVALID_ARG, "cns_host_alloc: null argument");
  *out = nullptr;
  HIPCHK(h, hipSetDevice(h->device));
  void* p = nullptr;
  HIPCHK(h, hipHostMalloc(&p, (size_t)std::max<uint64_t>(bytes, 1), hipHostMallocDefault));
  h->host_bufs.insert(p);
  *out = p;
  return CNS_OK;
}

int cns_host_free(cns_handle* h, void* p) {
  if (!h) return fail(h, CNS_ERR_INVALID_ARG, "cns_host_free: null handle");
  if (!p) return CNS_OK;
  auto it = h->host_bufs.find(p);
  if (it == h->host_bufs.end()) return fail(h, CNS_ERR_INVALID_ARG, "cns_host_free: not a buffer of cns_host_alloc on this handle");
  h->host_bufs.erase(it);
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipHostFree(p));
  return CNS_OK;
}

int cns_set_host_threads(cns_handle* h, uint32_t n) {
  if (!h) return fail(h, CNS_ERR_INVALID_ARG, "cns_set_host_threads: null handle");
  if (n > 64) return fail(h, CNS_ERR_INVALID_ARG, "cns_set_host_threads: more than 64 threads");
  h->host_threads = n;
  return CNS_OK;
}

int cns_select(cns_handle* h, int64_t now, const cns_job_soa* jobs, cns_placement_soa* out) {
  if (int rc = cns_upload_jobs(h, jobs)) return rc;
  if (int rc = cns_run_resident(h, now)) return rc;
  return cns_download(h, out);
}

// include/crane_gpu/preempt.h.  TryPreempt_ (JobScheduler.cpp:6378-6505) releases resources inside a cycle, which the
// pipelined kernels exclude by construction (node state monotone within a cycle: caches, predicted tiles, decoupled
// commits).  A cycle with preemption enabled therefore runs k_select with every job on its general path
// (KParams::general_only) and the device form of TryPreempt_ / PreemptSegTree between the res_total selection and the
// backfill (csrc/preempt_dev.inc).  Reservations are served (their virtual nodes carry their own job lists, cpp:6705), and
// so are partitions that share nodes (node-level job lists).
int cns_select_preempt(cns_handle* h, int64_t now, const cns_job_soa* jobs, const cns_preempt_soa* pre,
                       cns_placement_soa* out, cns_preempt_out* pout) {
  if (!h) return fail(h, CNS_ERR_INVALID_ARG, "cns_select_preempt: null handle");
  if (!pre || !pre->enabled) {
    if (int rc = cns_select(h, now, jobs, out)) return rc;
    if (pout) {
      if (pout->offsets && jobs) for (uint64_t j = 0; j <= jobs->num_jobs; ++j) pout->offsets[j] = 0;
      pout->num_cancelled = 0;
      const uint32_t n = pre ? std::min(pre->num_preempting, pout->preempting_capacity) : 0u;   // the set passes through
      for (uint32_t i = 0; i < n; ++i) pout->preempting_job_ids[i] = pre->preempting_job_ids[i];
      pout->num_preempting = n;
    }
    return CNS_OK;
  }
  if (!jobs || !out || !pout) return fail(h, CNS_ERR_INVALID_ARG, "cns_select_preempt: null argument");
  if (!h->have_nodes) return fail(h, CNS_ERR_STATE, "cns_select_preempt before cns_set_nodes");
  const u64 J = jobs->num_jobs;
  const u32 R = h->R;
  if (J && (!pre->pd_qos || !pre->pd_qos_priority || !pre->pd_priority)) return fail(h, CNS_ERR_INVALID_ARG, "cns_select_preempt: missing pending-job array");
  if (R && (!pre->rn_job_id || !pre->rn_qos || !pre->rn_qos_priority || !pre->rn_start_sec)) return fail(h, CNS_ERR_INVALID_ARG, "cns_select_preempt: missing running-job array");
  if (!pout->offsets || !pout->preempted || !pout->cancelled_job_ids || !pout->preempting_job_ids) return fail(h, CNS_ERR_INVALID_ARG, "cns_select_preempt: missing output array");
  HIPCHK(h, hipSetDevice(h->device));
  if (int rc = cns_upload_jobs(h, jobs)) return rc;
  const u32 A = (u32)h->ent_job.size();
  // m_preempting_set_: ids that no longer run are dropped, the others end at now + 1 (JobScheduler.cpp:6545-6559)
  std::map<u32, u32> id_to_rn;
  for (u32 r = 0; r < R; ++r) id_to_rn.emplace(pre->rn_job_id[r], r);
  std::vector<uint8_t> rj_pre(std::max<u32>(R, 1), 0);
  std::vector<u32> set_in;
  for (u32 i = 0; i < pre->num_preempting; ++i) {
    auto it = id_to_rn.find(pre->preempting_job_ids[i]);
    if (it == id_to_rn.end()) continue;
    rj_pre[it->second] = 1;
    set_in.push_back(pre->preempting_job_ids[i]);
  }
  std::vector<i64> ent_end = h->ent_end, rj_end(std::max<u32>(R, 1), 0), rj_start(std::max<u32>(R, 1), 0);
  std::vector<u32> rj_qos(std::max<u32>(R, 1), 0), rj_qprio(std::max<u32>(R, 1), 0), rj_off(R + 1, 0), rj_ent(std::max<u32>(A, 1), 0);
  for (u32 d = 0; d < A; ++d) rj_off[h->ent_job[d] + 1]++;
  for (u32 r = 0; r < R; ++r) rj_off[r + 1] += rj_off[r];
  {
    std::vector<u32> cur(rj_off.begin(), rj_off.end() - 1);
    for (u32 d = 0; d < A; ++d) rj_ent[cur[h->ent_job[d]]++] = d;
  }
  for (u32 r = 0; r < R; ++r) { rj_qos[r] = pre->rn_qos[r]; rj_qprio[r] = pre->rn_qos_priority[r]; rj_start[r] = pre->rn_start_sec[r]; }
  std::vector<char> have_end(std::max<u32>(R, 1), 0);
  for (u32 d = 0; d < A; ++d) {
    const u32 r = h->ent_job[d];
    if (rj_pre[r]) ent_end[d] = now + 1;
    rj_end[r] = std::max<i64>(ent_end[d], now + 1);   // :6513-6514
    have_end[r] = 1;
  }
  bool patched = false;
  for (u32 r = 0; r < R; ++r) patched = patched || rj_pre[r];
  // The device's running table carries the patched end times for THIS call only: whatever way the call ends, the resident
  // table goes back to the caller's end times (later cns_select / cns_run_resident calls reuse it).
  struct RestoreEnd {
    cns_engine* h; bool armed;
    ~RestoreEnd() {
      if (!armed) return;
      const std::string keep = h->err;   // the restore must not overwrite the call's own error
      if (upload(h, h->d_rn_end, h->ent_end) == CNS_OK) (void)hipStreamSynchronize(h->stream);
      h->err = keep;
    }
  } restore_end{h, false};
  if (patched) { restore_end.armed = true; if (int rc = upload(h, h->d_rn_end, ent_end)) return rc; }
  // No pending job's qos may preempt anything: TryPreempt_ returns at :6385 for every job, so the cycle is the plain one
  // (with the preempting jobs ending at now + 1) and runs on the pipelined kernels.
  bool any_list = false;
  for (u64 j = 0; j < J && !any_list; ++j) {
    const u32 q = pre->pd_qos[j];
    any_list = q < pre->num_qos && pre->qos_preempt_offsets[q + 1] > pre->qos_preempt_offsets[q];
  }
  if (!any_list) {
    int rc = cns_run_resident(h, now);
    if (rc) return rc;
    if (int rc3 = cns_download(h, out)) return rc3;
    for (u64 j = 0; j <= J; ++j) pout->offsets[j] = 0;
    pout->num_cancelled = 0;
    pout->num_preempting = 0;
    std::set<u32> keep(set_in.begin(), set_in.end());
    for (u32 id : keep) {
      if (pout->num_preempting >= pout->preempting_capacity) return fail(h, CNS_ERR_INVALID_ARG, "cns_select_preempt: preempting_capacity too small");
      pout->preempting_job_ids[pout->num_preempting++] = id;
    }
    return CNS_OK;
  }
  // qos preempt lists, pending-job fields (by queue index)
  std::vector<u32> qp_off(pre->num_qos + 1, 0), qp;
  for (u32 q = 0; q < pre->num_qos; ++q) {
    for (u32 i = pre->qos_preempt_offsets[q]; i < pre->qos_preempt_offsets[q + 1]; ++i) qp.push_back(pre->qos_preempt[i]);
    qp_off[q + 1] = (u32)qp.size();
  }
  if (qp.empty()) qp.push_back(0);
  std::vector<u32> pj_qos(std::max<u64>(J, 1), 0), pj_qprio(std::max<u64>(J, 1), 0);
  std::vector<double> pj_prio(std::max<u64>(J, 1), 0.0);
  for (u64 j = 0; j < J; ++j) { pj_qos[j] = pre->pd_qos[j]; pj_qprio[j] = pre->pd_qos_priority[j]; pj_prio[j] = pre->pd_priority[j]; }
  const u64 places = std::max<u64>(h->places, 1);
  // work buffers per partition (documented in include/crane_gpu/preempt.h): candidate / chosen lists sized for every running
  // job plus every pending job of the cycle (at most all of them hold resources on one job's nodes), bounded by 64 Mi entries
  // over all partitions; segment-tree pools of 65 536 nodes.  Exceeding either is CNS_ERR_UNSUPPORTED, not a device fault.
  const u32 pool_nodes = 1u << 16;
  const u32 cand_cap = (u32)std::max<u64>(4096, std::min<u64>((u64)R + J + 1, (64ull << 20) / std::max<u32>(h->P, 1)));
  const u32 out_cap = (u32)std::min<u64>(4 * (J + R) + 64, 1u << 28);
  DevBuf* B = h->d_pre;
  if (int rc = upload(h, B[B_QPOFF], qp_off)) return rc;
  if (int rc = upload(h, B[B_QP], qp)) return rc;
  if (int rc = upload(h, B[B_PJQOS], pj_qos)) return rc;
  if (int rc = upload(h, B[B_PJQP], pj_qprio)) return rc;
  if (int rc = upload(h, B[B_PJPRIO], pj_prio)) return rc;
  HIPCHK(h, B[B_PJREC0].ensure(std::max<u64>(J, 1) * 4)); HIPCHK(h, B[B_PJK].ensure(std::max<u64>(J, 1) * 4)); HIPCHK(h, B[B_PJEND].ensure(std::max<u64>(J, 1) * 8));
  { std::vector<u32> ej = h->ent_job, es = h->ent_slot; if (ej.empty()) { ej.push_back(0); es.push_back(0); }
    if (int rc = upload(h, B[B_RNJOB], ej)) return rc; if (int rc = upload(h, B[B_ENTSLOT], es)) return rc; }
  HIPCHK(h, B[B_ENTGONE].ensure(std::max<u32>(A, 1)));   // (zeroed at the start of every pass: run_resident_once)
  if (int rc = upload(h, B[B_RJQOS], rj_qos)) return rc;
  if (int rc = upload(h, B[B_RJQP], rj_qprio)) return rc;
  if (int rc = upload(h, B[B_RJSTART], rj_start)) return rc;
  if (int rc = upload(h, B[B_RJEND], rj_end)) return rc;
  if (int rc = upload(h, B[B_RJPRE], rj_pre)) return rc;
  if (int rc = upload(h, B[B_RJOFF], rj_off)) return rc;
  if (int rc = upload(h, B[B_RJENT], rj_ent)) return rc;
  HIPCHK(h, B[B_HEAD].ensure((size_t)std::max<u32>(h->S, 1) * 4));
  HIPCHK(h, B[B_RECNEXT].ensure(places * 4)); HIPCHK(h, B[B_RECORIG].ensure(places * 4)); HIPCHK(h, B[B_RECSLOT].ensure(places * 4));
  HIPCHK(h, B[B_RECGONE].ensure(places));
  // one block for: segment-tree pools | candidate lists | chosen lists | output counter | output pairs
  const size_t pool_b = (size_t)h->P * pool_nodes * sizeof(PreNode), cand_b = (size_t)h->P * cand_cap * 4;
  const size_t off_cand = align16(pool_b), off_chosen = off_cand + align16(cand_b), off_cnt = off_chosen + align16(cand_b), off_out = off_cnt + 16;
  HIPCHK(h, B[B_MISC].ensure(off_out + (size_t)out_cap * 8));
  char* misc = B[B_MISC].as<char>();
  PreParams& Q = h->pre_params;
  Q = PreParams{};
  Q.enabled = 1; Q.num_qos = pre->num_qos;
  Q.qp_off = B[B_QPOFF].as<u32>(); Q.qp = B[B_QP].as<u32>();
  Q.pj_qos = B[B_PJQOS].as<u32>(); Q.pj_qprio = B[B_PJQP].as<u32>(); Q.pj_prio = B[B_PJPRIO].as<double>();
  Q.pj_rec0 = B[B_PJREC0].as<u32>(); Q.pj_k = B[B_PJK].as<u32>(); Q.pj_end = B[B_PJEND].as<i64>();
  Q.rn_job = B[B_RNJOB].as<u32>(); Q.ent_slot = B[B_ENTSLOT].as<u32>(); Q.ent_gone = B[B_ENTGONE].as<uint8_t>();
  Q.rj_qos = B[B_RJQOS].as<u32>(); Q.rj_qprio = B[B_RJQP].as<u32>(); Q.rj_start = B[B_RJSTART].as<i64>(); Q.rj_end = B[B_RJEND].as<i64>();
  Q.rj_preempting = B[B_RJPRE].as<uint8_t>(); Q.rj_off = B[B_RJOFF].as<u32>(); Q.rj_ent = B[B_RJENT].as<u32>();
  Q.slot_head = B[B_HEAD].as<u32>(); Q.rec_next = B[B_RECNEXT].as<u32>(); Q.rec_orig = B[B_RECORIG].as<u32>();
  Q.rec_slot = B[B_RECSLOT].as<u32>(); Q.rec_gone = B[B_RECGONE].as<uint8_t>();
  Q.pool = misc; Q.pool_nodes = pool_nodes; Q.cand_cap = cand_cap;
  Q.cand = (u32*)(misc + off_cand); Q.chosen = (u32*)(misc + off_chosen);
  Q.out_cnt = (u32*)(misc + off_cnt); Q.out = (u32*)(misc + off_out); Q.out_cap = out_cap;
  // CNS_PREEMPT_TREE=literal: TryPreempt_ on the node-for-node trees only (else: the fall-back of a call that runs out of compressed records)
  { const char* tr = getenv("CNS_PREEMPT_TREE"); Q.literal_tree = (tr && !strcmp(tr, "literal")) ? 1u : ((tr && !strcmp(tr, "tiny")) ? 2u : 0u); }   // tiny: a handful of compressed records per call, the rest falls back
  // the partitions that have a pending job whose qos may preempt: only they run on k_select's general path (run_resident_once)
  h->pre_part.assign(h->P, 0);
  for (u64 j = 0; j < J; ++j) {
    const u32 q = pre->pd_qos[j];
    if (j < h->job_part.size() && h->job_part[(size_t)j] != kNone && q < pre->num_qos && pre->qos_preempt_offsets[q + 1] > pre->qos_preempt_offsets[q])
      h->pre_part[h->job_part[(size_t)j]] = 1;
  }
  h->pre_active = true;
  int rc = cns_run_resident(h, now);
  h->pre_active = false;
  if (rc) return rc;
  if (int rc3 = cns_download(h, out)) return rc3;
  // ---- the preempted lists: (pending job, reference) pairs in push_back order per job -> CSR by job ------------------
  u32 cnt = 0;
  HIPCHK(h, hipMemcpy(&cnt, misc + off_cnt, 4, hipMemcpyDeviceToHost));
  if (cnt > out_cap) return fail(h, CNS_ERR_UNSUPPORTED, "cns_select_preempt: more preemptions than the result buffer holds");
  std::vector<u32> pairs((size_t)cnt * 2 + 2);
  if (cnt) HIPCHK(h, hipMemcpy(pairs.data(), misc + off_out, (size_t)cnt * 8, hipMemcpyDeviceToHost));
  if (cnt > pout->capacity) return fail(h, CNS_ERR_INVALID_ARG, "cns_select_preempt: cns_preempt_out::capacity too small");
  for (u64 j = 0; j <= J; ++j) pout->offsets[j] = 0;
  for (u32 i = 0; i < cnt; ++i) pout->offsets[pairs[2 * i] + 1]++;
  for (u64 j = 0; j < J; ++j) pout->offsets[j + 1] += pout->offsets[j];
  {
    std::vector<u64> cur(pout->offsets, pout->offsets + J);
    for (u32 i = 0; i < cnt; ++i) pout->preempted[cur[pairs[2 * i]]++] = pairs[2 * i + 1];   // append order = push_back order within a job
  }
  // m_preempting_set_ / EnqueuePreemptCancel (:6786-6793), in queue order
  std::set<u32> pset(set_in.begin(), set_in.end());
  pout->num_cancelled = 0;
  for (u64 j = 0; j < J; ++j)
    for (u64 x = pout->offsets[j]; x < pout->offsets[j + 1]; ++x) {
      const u32 ref = pout->preempted[x];
      if (ref & CNS_PREEMPT_REF_PENDING) continue;
      const u32 id = pre->rn_job_id[ref];
      if (!pset.insert(id).second) continue;
      if (pout->num_cancelled >= pout->cancel_capacity) return fail(h, CNS_ERR_INVALID_ARG, "cns_select_preempt: cancel_capacity too small");
      pout->cancelled_job_ids[pout->num_cancelled++] = id;
    }
  pout->num_preempting = 0;
  for (u32 id : pset) {
    if (pout->num_preempting >= pout->preempting_capacity) return fail(h, CNS_ERR_INVALID_ARG, "cns_select_preempt: preempting_capacity too small");
    pout->preempting_job_ids[pout->num_preempting++] = id;
  }
  return CNS_OK;
}

int cns_get_partition_status(const cns_handle* h, uint8_t* status, uint32_t capacity) {
  if (!h || !status) return fail(const_cast<cns_handle*>(h), CNS_ERR_INVALID_ARG, "cns_get_partition_status: null argument");
  if (!h->have_nodes) return fail(const_cast<cns_handle*>(h), CNS_ERR_STATE, "cns_get_partition_status before cns_set_nodes");
  if (capacity < h->Pu) return fail(const_cast<cns_handle*>(h), CNS_ERR_INVALID_ARG, "cns_get_partition_status: capacity below the number of partitions");
  for (u32 p = 0; p < h->Pu; ++p) status[p] = h->upart_refused[p];
  return CNS_OK;
}

int cns_device_results(cns_handle* h, void** dptr, uint64_t* bytes) {
  if (!h || !dptr || !bytes) return fail(h, CNS_ERR_INVALID_ARG, "cns_device_results: null argument");
  if (!h->have_jobs) return fail(h, CNS_ERR_STATE, "cns_device_results before cns_upload_jobs");
  *dptr = h->d_results.p;
  *bytes = h->ro.total;
  return CNS_OK;
}

int cns_get_timing(const cns_handle* h, cns_timing* t) {
  if (!h || !t) return CNS_ERR_INVALID_ARG;
  *t = h->timing;
  return CNS_OK;
}

int cns_debug_get_costs(cns_handle* h, double* out) {
  if (!h || !out) return fail(h, CNS_ERR_INVALID_ARG, "cns_debug_get_costs: null argument");
  if (!h->have_run) return fail(h, CNS_ERR_STATE, "cns_debug_get_costs before a successful run");
  HIPCHK(h, hipSetDevice(h->device));
  std::vector<double> c(std::max<u32>(h->S, 1));
  HIPCHK(h, hipMemcpy(c.data(), h->d_cost.p, (size_t)h->S * sizeof(double), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < h->orig_pos_slot.size(); ++i) out[i] = h->orig_pos_slot[i] == kNone ? 0.0 : c[h->orig_pos_slot[i]];
  return CNS_OK;
}

const char* cns_debug_last_kernel(const cns_handle* h) { return h ? h->last_kernel.c_str() : ""; }

uint32_t cns_debug_engine_partitions(const cns_handle* h) { return (h && h->have_nodes) ? h->P : 0u; }

int cns_debug_get_prof(cns_handle* h, uint64_t* out, uint32_t capacity) {
  // cycle counters of the last run, 32 per partition; all zero unless the library was built with -DCNS_PROF
  if (!h || !out) return fail(h, CNS_ERR_INVALID_ARG, "cns_debug_get_prof: null argument");
#ifndef CNS_DEBUG_FLUSH_LOG   // (the diagnostics build also reads them after a run that failed)
  if (!h->have_run) return fail(h, CNS_ERR_STATE, "cns_debug_get_prof before a successful run");
#endif
  HIPCHK(h, hipSetDevice(h->device));
  // behind them (from index 32 * P): 16 always-on protocol counter slots per partition of k_wide (every build; wide_kernel.inc kWs*)
#ifdef CNS_DEBUG_FLUSH_LOG
  const size_t n = std::min<size_t>((size_t)h->P * (32 + 16) + 2048, capacity);   // + the flush log of partition 0 (diagnostics build)
#else
  const size_t n = std::min<size_t>((size_t)h->P * (32 + 16), capacity);
#endif
  HIPCHK(h, hipMemcpy(out, h->d_prof.p, n * sizeof(u64), hipMemcpyDeviceToHost));
  return CNS_OK;
}

int cns_debug_get_timeline(cns_handle* h, uint32_t node, uint32_t capacity, uint32_t* len, int64_t* t,
                           int64_t* cpu_raw, uint64_t* mem, uint64_t* core_lo, uint64_t* core_hi, uint64_t* gres) {
  if (!h || !len) return fail(h, CNS_ERR_INVALID_ARG, "cns_debug_get_timeline: null argument");
  if (!h->have_run) return fail(h, CNS_ERR_STATE, "cns_debug_get_timeline before a successful run");
  if (node >= h->N) return fail(h, CNS_ERR_INVALID_ARG, "cns_debug_get_timeline: node out of range");
  HIPCHK(h, hipSetDevice(h->device));
  const u32 slot = h->node_slot[node];
  if (slot == kNone) { *len = 0; return CNS_OK; }  // not schedulable / in no partition: no NodeState (cpp:6595)
  const char* blk = h->d_blocks.as<char>() + (size_t)slot * kBlockStride;
  NodeHdr hd;
  HIPCHK(h, hipMemcpy(&hd, blk, sizeof hd, hipMemcpyDeviceToHost));
  const u32 n = hd.len;
  *len = n;
  u32 m = std::min(n, capacity);
  std::vector<TlMem> e(std::max<u32>(m, 1));
  if (m) HIPCHK(h, hipMemcpy(e.data(), blk + sizeof(NodeHdr), (size_t)m * sizeof(TlMem), hipMemcpyDeviceToHost));
  for (u32 i = 0; i < m; ++i) {
    t[i] = e[i].t; cpu_raw[i] = e[i].cpu; mem[i] = e[i].mem; core_lo[i] = e[i].clo; core_hi[i] = e[i].chi; gres[i] = e[i].gres;
  }
  return CNS_OK;
}

int cns_debug_get_timeline_cores(cns_handle* h, uint32_t node, uint32_t capacity, uint64_t* core_w2, uint64_t* core_w3) {
  if (!h || !core_w2 || !core_w3) return fail(h, CNS_ERR_INVALID_ARG, "cns_debug_get_timeline_cores: null argument");
  if (!h->have_run) return fail(h, CNS_ERR_STATE, "cns_debug_get_timeline_cores before a successful run");
  if (node >= h->N) return fail(h, CNS_ERR_INVALID_ARG, "cns_debug_get_timeline_cores: node out of range");
  HIPCHK(h, hipSetDevice(h->device));
  const u32 slot = h->node_slot[node];
  if (slot == kNone) return CNS_OK;
  const char* blk = h->d_blocks.as<char>() + (size_t)slot * kBlockStride;
  NodeHdr hd;
  HIPCHK(h, hipMemcpy(&hd, blk, sizeof hd, hipMemcpyDeviceToHost));
  const u32 m = std::min(hd.len, capacity);
  for (u32 i = 0; i < m; ++i) core_w2[i] = core_w3[i] = 0;
  if (!h->wide_cores) return CNS_OK;   // (the TlExt array of a block is live only for snapshots with such core ids)
  std::vector<TlExt> e(std::max<u32>(m, 1));
  if (m) HIPCHK(h, hipMemcpy(e.data(), blk + sizeof(NodeHdr) + (size_t)kTlCap * sizeof(TlMem), (size_t)m * sizeof(TlExt), hipMemcpyDeviceToHost));
  for (u32 i = 0; i < m; ++i) { core_w2[i] = e[i].c2; core_w3[i] = e[i].c3; }
  return CNS_OK;
}

#include "priority_host.inc"
#include "limits_host.inc"
#include "steps_host.inc"

}  // extern "C"

#include "group_host.inc"
