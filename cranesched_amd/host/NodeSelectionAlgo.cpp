// GpuNodeSelectionAlgo: packs CraneCtld-shaped job / node objects into the SoA tables of the C ABI
// (include/crane_gpu/node_select.h), calls the engine and writes the placements back into
// PdJobInScheduler exactly where the reference's NodeSelect leaves them
// (src/CraneCtld/JobScheduler.cpp:6322-6331, :6772, :6768-6831; consumed at :1492-1600).
#include "NodeSelectionAlgo.h"
#include "../../include/crane_gpu/preempt.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <iterator>
#include <exception>
#include <thread>

#include "../../include/crane_gpu/node_select.h"
#include "../../include/crane_gpu/priority.h"
#include "../../include/crane_gpu/run_limits.h"
#include "../../include/crane_gpu/steps.h"

namespace crane {

namespace {
// (8 = CNS_REASON_ENGINE_REFUSED: not a reason of the reference — the job's partition lies outside the engine's limits, nothing was decided
// for it; RefusedJobs() lists exactly these jobs of the last cycle for the caller's CPU SchedulerAlgo)
const char* kReasonStr[] = {"", "Priority", "Resource", "Resource Reserved", "Partition Not Found", "", "Reservation Not Found", "Preempted", "GpuEngineRefused"};

// Page-locked storage (cns_host_alloc) for the arrays that cross the C ABI every cycle — the packed job table and the packed
// placements: kept across cycles (no page faults on a fresh array, no reallocation once they have grown to the queue's size) and
// copied by the DMA engines directly.  Each block carries its owner in a 64-byte header: without an engine (host-only benches, a
// failed cns_create) it is plain malloc memory.
struct PinCtx { cns_handle* h = nullptr; };
template <class T>
struct PinAlloc {
  using value_type = T;
  using propagate_on_container_move_assignment = std::true_type;   // `v = PinVec<T>()` hands v's block back at once (the destructor relies on it)
  using propagate_on_container_swap = std::true_type;
  PinCtx* ctx = nullptr;
  PinAlloc() = default;
  explicit PinAlloc(PinCtx* c) : ctx(c) {}
  template <class U> PinAlloc(const PinAlloc<U>& o) : ctx(o.ctx) {}
  T* allocate(size_t n) {
    const size_t bytes = n * sizeof(T) + 64;
    void* base = nullptr;
    cns_handle* owner = nullptr;
    if (ctx && ctx->h && cns_host_alloc(ctx->h, bytes, &base) == 0 && base) owner = ctx->h;
    else base = std::malloc(bytes);
    if (!base) throw std::bad_alloc();
    *reinterpret_cast<cns_handle**>(base) = owner;
    return reinterpret_cast<T*>(static_cast<char*>(base) + 64);
  }
  void deallocate(T* p, size_t) noexcept {
    char* base = reinterpret_cast<char*>(p) - 64;
    cns_handle* owner = *reinterpret_cast<cns_handle**>(base);
    if (owner) (void)cns_host_free(owner, base);
    else std::free(base);
  }
  template <class U> bool operator==(const PinAlloc<U>& o) const { return ctx == o.ctx; }
  template <class U> bool operator!=(const PinAlloc<U>& o) const { return ctx != o.ctx; }
};
template <class T> using PinVec = std::vector<T, PinAlloc<T>>;
}

struct GpuNodeSelectionAlgo::Impl {
  cns_handle* h = nullptr;     // the engine (several devices: the first device's, for page-locked buffers and the kernels either side of the path)
  cns_group* grp = nullptr;    // several devices (include/crane_gpu/node_select.h "several devices"): snapshot, running set and cycle go through it
  std::vector<int> devices;
  // dense indices of the current snapshot
  std::vector<CranedId> node_name;
  std::vector<uint64_t> node_mem_sw;   // res_total.memory_sw_bytes per node (what an exclusive job is allocated)
  std::unordered_map<CranedId, uint32_t> node_idx;
  std::unordered_map<PartitionId, uint32_t> part_idx;
  std::unordered_map<std::string, uint32_t> resv_idx;
  // (name, type) -> class; slot path -> bit, per class in lexicographic path order (std::set<SlotId> order)
  std::vector<std::pair<std::string, std::string>> classes;
  std::map<std::string, uint32_t> name_id;
  std::vector<std::map<SlotId, uint32_t>> class_slot_bit;  // per class: slot path -> absolute bit
  std::vector<std::vector<SlotId>> class_bit_slot;          // per class: bit offset -> slot path
  cns_gres_layout layout{};
  bool have_snapshot = false;
  std::vector<const PdJobInScheduler*> last_ord;                      // the jobs of the last cns_select, in its order
  std::unordered_map<const PdJobInScheduler*, uint64_t> last_index;  // job -> its index there (built when the run-limit pass asks)
  // Incremental packing of the running jobs: an allocation never changes while its job runs, so its
  // dense form (node indices, core / GRES masks) is kept per job id across cycles; a cycle costs one lookup per
  // running job instead of one string lookup + set -> mask conversion per allocated node.  Entries of jobs that did
  // not show up in a cycle are dropped; a new snapshot (new dense indices) drops everything.
  struct AllocRec { uint32_t node; int64_t cpu; uint64_t mem, lo, hi, g, w2, w3; bool ovf = false; };   // lo / hi / w2 / w3: core ids 0..255; ovf: it also holds an id >= 256
  struct PackedAlloc {
    uint32_t resv;
    uint64_t gen;
    std::vector<AllocRec> recs;
  };
  // the packed node table / reservations of the current snapshot, kept so that a state flip of one craned
  // (CranedUp / CranedDown / drain) re-sends them without touching a string
  std::vector<int64_t> n_cpu, v_start, v_end, v_cpu;
  std::vector<uint64_t> n_mem, n_lo, n_hi, n_gres, v_mem, v_lo, v_hi, v_g, n_w2, n_w3, v_w2, v_w3;
  std::vector<uint8_t> n_sched;
  std::vector<uint8_t> n_unsup;   // node -> it cannot be expressed in the ABI's formats (a core id >= 256, GRES slots / classes beyond the 64-bit
                                  // mask): the engine refuses the partitions that list it (cns_node_soa::unsupported) and serves the others
  std::vector<uint32_t> n_poff, n_pnodes, v_off, v_node;
  int push_tables(std::string& err) {   // cns_set_nodes + cns_set_reservations from the packed arrays
    cns_node_soa nd{};
    nd.num_nodes = (uint32_t)n_cpu.size();
    nd.num_partitions = (uint32_t)n_poff.size() - 1;
    nd.cpu_total_raw = n_cpu.data(); nd.mem_total = n_mem.data(); nd.core_lo = n_lo.data(); nd.core_hi = n_hi.data();
    nd.gres_slots = n_gres.data(); nd.schedulable = n_sched.data();
    nd.part_offsets = n_poff.data(); nd.part_nodes = n_pnodes.data();
    nd.gres = layout;
    nd.core_w2 = n_w2.data(); nd.core_w3 = n_w3.data();
    nd.unsupported = n_unsup.empty() ? nullptr : n_unsup.data();
    int st = grp ? cns_group_set_nodes(grp, &nd) : cns_set_nodes(h, &nd);
    if (st != 0) { err = grp ? cns_group_last_error(grp) : cns_last_error(h); return st; }
    if (!v_start.empty()) {
      cns_resv_soa rv{};
      rv.num_resv = (uint32_t)v_start.size(); rv.num_allocs = (uint32_t)v_node.size();
      rv.start_sec = v_start.data(); rv.end_sec = v_end.data(); rv.alloc_offsets = v_off.data(); rv.alloc_node = v_node.data();
      rv.alloc_cpu_raw = v_cpu.data(); rv.alloc_mem = v_mem.data(); rv.alloc_core_lo = v_lo.data(); rv.alloc_core_hi = v_hi.data();
      rv.alloc_gres = v_g.data(); rv.alloc_core_w2 = v_w2.data(); rv.alloc_core_w3 = v_w3.data();
      st = grp ? cns_group_set_reservations(grp, &rv) : cns_set_reservations(h, &rv);
      if (st != 0) { err = grp ? cns_group_last_error(grp) : cns_last_error(h); return st; }
    }
    return 0;
  }
  std::unordered_map<job_id_t, PackedAlloc> alloc_cache;
  // preemption (include/crane_gpu/preempt.h)
  bool preempt_enabled = false;
  std::unordered_map<std::string, uint32_t> qos_id;                 // qos name -> dense id
  std::vector<std::vector<uint32_t>> qos_preempt;                   // id -> ids it may preempt
  std::set<job_id_t> preempting;                                    // m_preempting_set_
  std::vector<job_id_t> cancelled;                                  // EnqueuePreemptCancel of the last cycle
  std::vector<RnJobInScheduler*> r_src;                             // packed running job r -> the caller's object
  bool r_src_valid = false;                                         // r_src was filled by THIS pack (pack_running), not by an earlier cycle
  uint32_t qos_of(const std::string& name) {
    auto it = qos_id.find(name);
    if (it != qos_id.end()) return it->second;
    const uint32_t id = (uint32_t)qos_id.size();
    qos_id.emplace(name, id);
    qos_preempt.emplace_back();   // a qos the table does not know: nothing to preempt (cpp:6537-6539)
    return id;
  }
  uint64_t alloc_gen = 0;
  bool use_alloc_cache = true;
  std::vector<int64_t> r_end, r_cpu;
  std::vector<uint32_t> r_off, r_node, r_resv;
  std::vector<uint64_t> r_mem, r_lo, r_hi, r_g, r_w2, r_w3;


  // ---- pending jobs -> cns_job_soa arrays, and placements -> PdJobInScheduler.  The reference's structures — a
  // std::set of core ids, string-keyed maps per job — are what makes the write-back expensive (~0.65 us per placed
  // job); a caller that can consume the SoA of the C ABI directly skips it. ------------------------------------------
  // Host threads (SetHostThreads; default 1): the loops are per job and independent, but the write-back is bound by the allocator —
  // eight small allocations per placed job — and by the page faults behind it: on the MI355X box's host 1 M jobs take 440 ms on one
  // thread, 261 ms on 4, 445 ms on 16 (pack: 51 / 37 / 31 ms; profiles/r03_host_threads.txt).  The deferred write-back (40 ms) is
  // the better lever.
  int host_threads = 1;
  template <class F>
  void parallel_for(size_t n, F&& body) const {
    const size_t T = (size_t)std::max(1, host_threads);
    if (T == 1 || n < 4096) { body((size_t)0, n); return; }
    // an exception on a worker (bad_alloc: the write-back is allocator-bound) is carried to the caller after every thread has
    // been joined — a thread that lets one escape, or a joinable thread that is destroyed, would terminate CraneCtld
    std::vector<std::thread> th;
    std::vector<std::exception_ptr> err(T);
    struct Joiner { std::vector<std::thread>& t; ~Joiner() { for (auto& x : t) if (x.joinable()) x.join(); } } joiner{th};
    const size_t chunk = (n + T - 1) / T;
    auto guarded = [&body, &err](size_t slot, size_t a, size_t b) {
      try { body(a, b); } catch (...) { err[slot] = std::current_exception(); }
    };
    for (size_t t = 1; t < T; ++t) {
      const size_t a = std::min(n, t * chunk), b = std::min(n, a + chunk);
      if (a < b) th.emplace_back(guarded, t, a, b);
    }
    guarded(0, (size_t)0, std::min(n, chunk));
    for (auto& x : th) x.join();
    for (auto& e : err) if (e) std::rethrow_exception(e);
  }
  PinCtx pin;   // (pin.h = h once the engine exists)
  struct PackedJobs {
    PinVec<uint32_t> part, k, nt, tmin, tmax, inodes, enodes, jresv;
    PinVec<int64_t> L, ncpu, tcpu;
    PinVec<uint64_t> nmem, tmem, ioff, eoff;
    PinVec<uint8_t> excl, skip, gtot, gspec;
    explicit PackedJobs(PinCtx* c = nullptr)
        : part(PinAlloc<uint32_t>(c)), k(PinAlloc<uint32_t>(c)), nt(PinAlloc<uint32_t>(c)), tmin(PinAlloc<uint32_t>(c)), tmax(PinAlloc<uint32_t>(c)),
          inodes(PinAlloc<uint32_t>(c)), enodes(PinAlloc<uint32_t>(c)), jresv(PinAlloc<uint32_t>(c)), L(PinAlloc<int64_t>(c)), ncpu(PinAlloc<int64_t>(c)),
          tcpu(PinAlloc<int64_t>(c)), nmem(PinAlloc<uint64_t>(c)), tmem(PinAlloc<uint64_t>(c)), ioff(PinAlloc<uint64_t>(c)), eoff(PinAlloc<uint64_t>(c)),
          excl(PinAlloc<uint8_t>(c)), skip(PinAlloc<uint8_t>(c)), gtot(PinAlloc<uint8_t>(c)), gspec(PinAlloc<uint8_t>(c)) {}
  };
  PackedJobs packed{&pin};   // the job table of the cycle: page-locked, reused
  double t_pack_ms = 0, t_engine_ms = 0, t_write_ms = 0;   // the last NodeSelect: pack | cns_select (H2D + kernels + D2H) | write-back
  struct PlacementStore;
  // (round 5: everything per job happens in ONE parallel pass — the arrays are sized, not zero-filled (every element is written below), the
  // include / exclude lists are counted there and only walked again when some job has one, and the write-back's per-job side table is filled
  // there too: the serial passes over 1 M job objects and the 150 MB of memsets were 30 of the 44 ms a cycle's packing took)
  // (`places`: the sum of node_num over the queue, and the jobs' preempted_jobs cleared — NodeSelect's two other walks over the 1 M job objects,
  // 10 ms on one thread for cache misses alone, done where the pass has the object in hand)
  void pack_pending(const std::vector<PdJobInScheduler*>& ord, PackedJobs& B, PlacementStore* S = nullptr, uint64_t* places = nullptr) const {
    const size_t J = ord.size();
    B.part.resize(J); B.k.resize(J); B.nt.resize(J); B.tmin.resize(J); B.tmax.resize(J);
    B.L.resize(J); B.ncpu.resize(J); B.tcpu.resize(J); B.nmem.resize(J); B.tmem.resize(J);
    B.excl.resize(J); B.skip.resize(J); B.gtot.resize(J * CNS_MAX_GRES_NAMES); B.gspec.resize(J * CNS_MAX_GRES_CLASSES);
    B.jresv.resize(J);
    B.ioff.resize(J + 1); B.eoff.resize(J + 1);
    if (S) { S->excl.resize(J); S->msw_node.resize(J); S->msw_task.resize(J); }
    std::atomic<uint64_t> n_lists{0}, n_places{0};
    parallel_for(J, [&](size_t a, size_t b) {
      uint64_t lists = 0, plc = 0;
      for (size_t j = a; j < b; ++j) {
        if (j + 8 < b) __builtin_prefetch(ord[j + 8]);
        const PdJobInScheduler& p = *ord[j];
        if (places) { plc += p.node_num; if (!p.preempted_jobs.empty()) ord[j]->preempted_jobs.clear(); }
        memset(&B.gtot[j * CNS_MAX_GRES_NAMES], 0, CNS_MAX_GRES_NAMES);
        memset(&B.gspec[j * CNS_MAX_GRES_CLASSES], 0, CNS_MAX_GRES_CLASSES);
        B.jresv[j] = CNS_RESV_NONE;
        B.ioff[j + 1] = p.included_nodes.size(); B.eoff[j + 1] = p.excluded_nodes.size();   // (counts; turned into offsets below)
        lists += B.ioff[j + 1] + B.eoff[j + 1];
        if (S) { S->excl[j] = p.exclusive; S->msw_node[j] = p.req_node_res_view.memory_sw_bytes; S->msw_task[j] = p.req_task_res_view.memory_sw_bytes; }
        auto pit = part_idx.find(p.partition_id);
        B.part[j] = pit == part_idx.end() ? 0xFFFFFFFFu : pit->second;  // -> "Partition Not Found" (cpp:6748-6752)
        B.L[j] = p.time_limit;
        B.ncpu[j] = p.req_node_res_view.cpu_count.raw;
        B.nmem[j] = p.req_node_res_view.memory_bytes;
        B.tcpu[j] = p.req_task_res_view.cpu_count.raw;
        B.tmem[j] = p.req_task_res_view.memory_bytes;
        B.k[j] = p.node_num; B.nt[j] = p.ntasks; B.tmin[j] = p.ntasks_per_node_min; B.tmax[j] = p.ntasks_per_node_max;
        B.excl[j] = p.exclusive;
        B.skip[j] = !p.reason.empty();  // cpp:6744
        if (!p.reservation.empty()) {  // scheduled by the reservation's scheduler (cpp:6754-6760); unknown -> "Reservation Not Found"
          auto rit = resv_idx.find(p.reservation);
          B.jresv[j] = rit == resv_idx.end() ? 0xFFFFFFFEu : rit->second;
        }
        for (const auto& [name, gc] : p.req_node_res_view.gres_map) {
          auto nit = name_id.find(name);
          uint64_t tot = gc.total;
          if (nit == name_id.end()) { if (tot || !gc.specified.empty()) B.gtot[j * CNS_MAX_GRES_NAMES] = 255; continue; }  // name absent everywhere: never fits
          B.gtot[j * CNS_MAX_GRES_NAMES + nit->second] = (uint8_t)std::min<uint64_t>(tot, 255);
          for (const auto& [type, cnt] : gc.specified) {
            int c = class_of(name, type);
            if (c < 0) { if (cnt) B.gtot[j * CNS_MAX_GRES_NAMES + nit->second] = 255; continue; }               // type absent everywhere
            B.gspec[j * CNS_MAX_GRES_CLASSES + c] = (uint8_t)std::min<uint64_t>(cnt, 127);
          }
        }
      }
      n_lists.fetch_add(lists, std::memory_order_relaxed);
      n_places.fetch_add(plc, std::memory_order_relaxed);
    });
    if (places) *places = n_places.load();
    // include / exclude lists (rare): CSR, in order
    B.ioff[0] = 0; B.eoff[0] = 0; B.inodes.clear(); B.enodes.clear();
    if (n_lists.load() == 0) {
      // (every count written by the pass is zero: they ARE the offsets)
    } else {
      for (size_t j = 0; j < J; ++j) {
        const PdJobInScheduler& p = *ord[j];
        for (const auto& n : p.included_nodes) { auto it = node_idx.find(n); B.inodes.push_back(it == node_idx.end() ? 0xFFFFFFFEu : it->second); }
        B.ioff[j + 1] = B.inodes.size();
        for (const auto& n : p.excluded_nodes) { auto it = node_idx.find(n); if (it != node_idx.end()) B.enodes.push_back(it->second); }
        B.eoff[j + 1] = B.enodes.size();
      }
    }
    if (B.inodes.empty()) B.inodes.push_back(0);
    if (B.enodes.empty()) B.enodes.push_back(0);
  }
  // what JobScheduler.cpp:1492-1600 consumes
  // lazy (default): a job that did not start now leaves NodeSelect with a pending reason, and the commit loop then reads
  // nothing but that reason, start_time and priority (JobScheduler.cpp:1503-1510) — its node list and ResourceInNodeV3
  // objects are not materialised (3.3 us per job at 1 M jobs: more than half a GPU cycle, profiles/r01_host_pack_bench.txt)
  bool lazy_write_back = true;
  // ... deferred: a job that starts now gets reason, start / end time and craned_ids — what the commit loop reads before the
  // run-limit admission (:1503-1573) —, its craned_id_to_task_num / allocated_res when the caller asks (MaterializeAllocation: the
  // jobs it launches, :1590-1600).  A 1 M-job cycle admits a third of what it starts (C4's limits), the product orders 100 k per cycle.
  bool deferred_write_back = false;
  void write_back(const std::vector<PdJobInScheduler*>& ord, const cns_placement_soa& o) const {
    parallel_for(ord.size(), [&](size_t a, size_t b) {
      for (size_t j = a; j < b; ++j) {
        PdJobInScheduler& p = *ord[j];
        const uint8_t r = o.reason[j];
        if (r == CNS_REASON_SKIPPED) continue;  // the caller's reason stays
        p.reason = kReasonStr[r];
        p.craned_ids.clear(); p.craned_id_to_task_num.clear(); p.allocated_res.clear();
        if (o.start_sec[j] == 0) continue;      // nothing placed ("Resource" / "Partition Not Found" / "Priority" beyond the batch)
        p.start_time = o.start_sec[j];
        p.end_time = p.start_time + p.time_limit;  // cpp:6772
        if (lazy_write_back && r != CNS_REASON_NONE) continue;   // backfilled for later: only reason + start time are consumed
        for (uint64_t q = o.place_offsets[j]; q < o.place_offsets[j + 1]; ++q) {
          if (o.node_idx[q] == CNS_NODE_NONE) continue;
          const CranedId& cid = node_name[o.node_idx[q]];
          p.craned_ids.push_back(cid);
          if (deferred_write_back) continue;
          p.craned_id_to_task_num[cid] = o.ntasks[q];
          ResourceInNodeV3& res = p.allocated_res[cid];   // built in place (the map was cleared above)
          fill_res(res, o.cpu_raw[q], o.mem[q], o.core_lo[q], o.core_hi[q], o.gres[q], o.core_w2 ? o.core_w2[q] : 0, o.core_w3 ? o.core_w3[q] : 0);
          // an exclusive job is allocated the node's whole res_total, memory_sw_bytes included (JobScheduler.cpp:6309-6310)
          res.memory_sw_bytes = p.exclusive ? node_mem_sw[o.node_idx[q]]
                                            : p.req_node_res_view.memory_sw_bytes + p.req_task_res_view.memory_sw_bytes * o.ntasks[q];
        }
      }
    });
  }

  // ---- event-fed mirror: job id -> what MallocResourceFromNode was told, packed when it was told -----------------------
  struct MirrorRec { CranedId craned; ResourceInNodeV3 res; AllocRec packed; bool known; };
  struct MirrorJob { TimeSec end_time = 0; std::string resv; std::vector<MirrorRec> recs; };
  std::map<job_id_t, MirrorJob> mirror;   // ascending job id = the order of the reference's running-job map
  // The packed form of the mirror is kept between cycles and PATCHED: jobs that ended are squeezed out in one sequential pass, jobs
  // that started (ids above everything packed) are appended, a changed end time is written in place.  Anything else — an allocation
  // added to or taken from a job that is already packed and still runs, an id below the packed range, a reservation that changes,
  // a new snapshot, an explicit running vector in between — falls back to the full walk over the map.
  std::vector<job_id_t> r_job;            // packed job ids, ascending (valid while mirror_packed)
  bool mirror_packed = false;             // r_* hold a packed mirror
  bool mirror_full = true;                // the next pack walks the whole map
  job_id_t m_last_id = 0;                 // highest packed id
  std::vector<job_id_t> m_removed, m_touched, m_info;   // events since the last pack
  size_t mirror_full_packs = 0, mirror_patch_packs = 0;
  void pack_rec(MirrorRec& r) {
    if (r.packed.ovf) { --mirror_ovf; r.packed.ovf = false; }
    auto it = node_idx.find(r.craned);
    r.known = it != node_idx.end();
    if (!r.known) return;
    r.packed.node = it->second;
    r.packed.cpu = r.res.cpu_set.cpu_count.raw;
    r.packed.mem = r.res.memory_bytes;
    // (an id >= 256 on a node the snapshot flags unsupported is no overflow of the cycle: that node's partitions are refused as a group,
    // and the engine drops allocations on nodes it schedules for nobody)
    r.packed.ovf = core_masks(r.res.cpu_set.core_ids, r.packed.lo, r.packed.hi, r.packed.w2, r.packed.w3) && !(r.packed.node < n_unsup.size() && n_unsup[r.packed.node]);
    if (r.packed.ovf) ++mirror_ovf;
    r.packed.g = gres_mask(r.res.gres);
  }
  void repack_mirror() {   // after a new snapshot
    for (auto& [id, mj] : mirror)
      for (auto& r : mj.recs) pack_rec(r);
    mirror_full = true;
  }
  void append_mirror_job(job_id_t id, const MirrorJob& mj) {
    uint32_t rv = CNS_RESV_NONE;
    if (!mj.resv.empty()) {
      auto it = resv_idx.find(mj.resv);
      if (it == resv_idx.end()) return;
      rv = it->second;
    }
    r_job.push_back(id);
    r_resv.push_back(rv);
    r_end.push_back(mj.end_time);
    for (const MirrorRec& r : mj.recs) {
      if (!r.known) continue;
      const AllocRec& a = r.packed;
      r_node.push_back(a.node); r_cpu.push_back(a.cpu); r_mem.push_back(a.mem);
      r_lo.push_back(a.lo); r_hi.push_back(a.hi); r_g.push_back(a.g); r_w2.push_back(a.w2); r_w3.push_back(a.w3);
    }
    r_off.push_back((uint32_t)r_node.size());
  }
  void pack_from_mirror_full() {
    r_end.clear(); r_cpu.clear(); r_node.clear(); r_resv.clear(); r_mem.clear(); r_lo.clear(); r_hi.clear(); r_g.clear(); r_w2.clear(); r_w3.clear();
    r_job.clear();
    r_off.assign(1, 0);
    for (const auto& [id, mj] : mirror) append_mirror_job(id, mj);
    m_last_id = mirror.empty() ? 0 : mirror.rbegin()->first;
    ++mirror_full_packs;
  }
  // false: something happened that the patch does not cover
  bool pack_from_mirror_patch() {
    for (job_id_t id : m_touched) if (mirror.count(id)) return false;   // (a touched job that ended since is just a removal)
    for (job_id_t id : m_info) {
      auto mit = mirror.find(id);
      if (mit == mirror.end()) continue;
      auto pos = std::lower_bound(r_job.begin(), r_job.end(), id);
      if (pos == r_job.end() || *pos != id) return false;               // not packed (its reservation was unknown): the full walk decides
      const size_t j = (size_t)(pos - r_job.begin());
      uint32_t rv = CNS_RESV_NONE;
      if (!mit->second.resv.empty()) {
        auto it = resv_idx.find(mit->second.resv);
        if (it == resv_idx.end()) return false;
        rv = it->second;
      }
      if (rv != r_resv[j]) return false;
      r_end[j] = mit->second.end_time;
    }
    if (!m_removed.empty()) {   // squeeze the ended jobs out: one sequential pass, runs of kept jobs move as blocks
      std::sort(m_removed.begin(), m_removed.end());
      m_removed.erase(std::unique(m_removed.begin(), m_removed.end()), m_removed.end());
      const size_t Jn = r_job.size();
      size_t wj = 0, ri = 0;          // next job slot to write, next removed id to meet
      uint32_t wa = 0;                // next record slot to write
      size_t j = 0;
      while (j < Jn) {
        while (ri < m_removed.size() && m_removed[ri] < r_job[j]) ++ri;
        if (ri < m_removed.size() && m_removed[ri] == r_job[j]) { ++j; continue; }
        size_t e = j + 1;             // the run of kept jobs [j, e)
        while (e < Jn) {
          while (ri < m_removed.size() && m_removed[ri] < r_job[e]) ++ri;
          if (ri < m_removed.size() && m_removed[ri] == r_job[e]) break;
          ++e;
        }
        const uint32_t a0 = r_off[j], a1 = r_off[e];
        if (wj != j) {
          std::move(r_job.begin() + j, r_job.begin() + e, r_job.begin() + wj);
          std::move(r_end.begin() + j, r_end.begin() + e, r_end.begin() + wj);
          std::move(r_resv.begin() + j, r_resv.begin() + e, r_resv.begin() + wj);
          std::move(r_node.begin() + a0, r_node.begin() + a1, r_node.begin() + wa);
          std::move(r_cpu.begin() + a0, r_cpu.begin() + a1, r_cpu.begin() + wa);
          std::move(r_mem.begin() + a0, r_mem.begin() + a1, r_mem.begin() + wa);
          std::move(r_lo.begin() + a0, r_lo.begin() + a1, r_lo.begin() + wa);
          std::move(r_hi.begin() + a0, r_hi.begin() + a1, r_hi.begin() + wa);
          std::move(r_g.begin() + a0, r_g.begin() + a1, r_g.begin() + wa);
          std::move(r_w2.begin() + a0, r_w2.begin() + a1, r_w2.begin() + wa);
          std::move(r_w3.begin() + a0, r_w3.begin() + a1, r_w3.begin() + wa);
        }
        const uint32_t shift = a0 - wa;   // (offsets of the run, ascending: r_off[wj + 1 ..] are behind what is still to be read)
        for (size_t x = j; x < e; ++x) r_off[wj + (x - j) + 1] = r_off[x + 1] - shift;
        wj += e - j; wa += a1 - a0;
        j = e;
      }
      r_job.resize(wj); r_end.resize(wj); r_resv.resize(wj); r_off.resize(wj + 1);
      r_node.resize(wa); r_cpu.resize(wa); r_mem.resize(wa); r_lo.resize(wa); r_hi.resize(wa); r_g.resize(wa); r_w2.resize(wa); r_w3.resize(wa);
    }
    for (auto it = mirror.upper_bound(m_last_id); it != mirror.end(); ++it) append_mirror_job(it->first, it->second);
    if (!mirror.empty()) m_last_id = std::max(m_last_id, mirror.rbegin()->first);
    ++mirror_patch_packs;
    return true;
  }
  void pack_from_mirror() {
    r_src.clear();        // the mirror has no RnJobInScheduler objects: a cycle with preemption must be refused, never served from an
    r_src_valid = false;  // earlier explicit cycle's (freed) pointers that happen to match in number
    if (!mirror_packed || mirror_full || !pack_from_mirror_patch()) pack_from_mirror_full();
    packed_from_mirror = true;
    mirror_packed = true;
    mirror_full = false;
    m_removed.clear(); m_touched.clear(); m_info.clear();
  }
  // checksum of the packed running arrays that ignores the order of a job's per-node records
  uint64_t running_checksum_canonical() const {
    auto h64 = [](uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; return x ^ (x >> 33); };
    uint64_t acc = 1469598103934665603ull;
    for (size_t j = 0; j + 1 < r_off.size(); ++j) {
      uint64_t s = h64((uint64_t)r_end[j]) ^ h64(r_resv[j] + 0x9e37ull);
      for (uint32_t a = r_off[j]; a < r_off[j + 1]; ++a)
        s += h64(h64(r_node[a]) ^ h64((uint64_t)r_cpu[a] + 1) ^ h64(r_mem[a] + 2) ^ h64(r_lo[a] + 3) ^ h64(r_hi[a] + 4) ^ h64(r_g[a] + 5) ^
                 ((r_w2[a] | r_w3[a]) ? h64(r_w2[a] + 6) ^ h64(r_w3[a] + 7) : 0));
      acc = (acc ^ h64(s)) * 1099511628211ull;
    }
    return acc;
  }

  // running jobs -> cns_running_soa arrays (JobScheduler.cpp:6681-6709); order = the caller's vector
  void pack_running(const std::vector<std::unique_ptr<RnJobInScheduler>>& running_jobs) {
    r_end.clear(); r_cpu.clear(); r_node.clear(); r_resv.clear(); r_mem.clear(); r_lo.clear(); r_hi.clear(); r_g.clear(); r_w2.clear(); r_w3.clear();
    r_src.clear();
    r_src_valid = true;
    mirror_packed = false;   // (r_* now hold the caller's vector)
    packed_from_mirror = false;
    run_overflow = false;
    r_off.assign(1, 0);
    ++alloc_gen;
    PackedAlloc scratch;
    for (const auto& rn : running_jobs) {
      uint32_t rv = CNS_RESV_NONE;
      if (!rn->reservation.empty()) {  // allocated inside the reservation's own node states (cpp:6692-6707)
        auto it = resv_idx.find(rn->reservation);
        if (it == resv_idx.end()) continue;
        rv = it->second;
      }
      PackedAlloc* pa = nullptr;
      if (use_alloc_cache) {
        auto it = alloc_cache.find(rn->job_id);
        if (it != alloc_cache.end() && it->second.resv == rv) pa = &it->second;
      }
      if (!pa) {
        PackedAlloc& d = use_alloc_cache ? alloc_cache[rn->job_id] : scratch;
        d.resv = rv;
        d.recs.clear();
        for (const auto& [cid, res] : rn->allocated_res) {
          auto it = node_idx.find(cid);
          if (it == node_idx.end()) continue;
          AllocRec a;
          a.node = it->second;
          a.cpu = res.cpu_set.cpu_count.raw;
          a.mem = res.memory_bytes;
          a.ovf = core_masks(res.cpu_set.core_ids, a.lo, a.hi, a.w2, a.w3) && !(a.node < n_unsup.size() && n_unsup[a.node]);
          a.g = gres_mask(res.gres);
          d.recs.push_back(a);
        }
        pa = &d;
      }
      pa->gen = alloc_gen;
      r_resv.push_back(rv);
      r_end.push_back(rn->end_time);
      r_src.push_back(rn.get());
      for (const AllocRec& a : pa->recs) {
        run_overflow = run_overflow || a.ovf;
        r_node.push_back(a.node); r_cpu.push_back(a.cpu); r_mem.push_back(a.mem);
        r_lo.push_back(a.lo); r_hi.push_back(a.hi); r_g.push_back(a.g); r_w2.push_back(a.w2); r_w3.push_back(a.w3);
      }
      r_off.push_back((uint32_t)r_node.size());
    }
    if (use_alloc_cache && alloc_cache.size() > r_end.size())  // jobs that ended since the last cycle
      for (auto it = alloc_cache.begin(); it != alloc_cache.end();) it = it->second.gen != alloc_gen ? alloc_cache.erase(it) : std::next(it);
  }

  int class_of(const std::string& name, const std::string& type) const {
    for (size_t c = 0; c < classes.size(); ++c)
      if (classes[c].first == name && classes[c].second == type) return (int)c;
    return -1;
  }
  uint64_t gres_mask(const DedicatedResourceInNode& d) const {
    uint64_t m = 0;
    for (const auto& [name, tm] : d)
      for (const auto& [type, slots] : tm) {
        int c = class_of(name, type);
        if (c < 0) continue;
        for (const auto& s : slots) {
          auto it = class_slot_bit[c].find(s);
          if (it != class_slot_bit[c].end()) m |= 1ull << it->second;
        }
      }
    return m;
  }
  // core ids as four 64-bit masks (ids 0..255, ABI 3); an id >= 256 does not fit the engine's model: it is RECORDED (core_overflow) and the
  // snapshot / cycle is refused — silently dropping it would make integer requests fail the `popc < n` test, or come back
  // without core ids, where ResourceView::GetFeasibleResourceInNode (PublicHeader.cpp:528-538) fits them
  // The overflow is a property of the RECORD it was found in, kept with the record: a cached or mirrored allocation that holds such an
  // id refuses every cycle it is part of (not only the cycle that first packed it), and a step pass that meets one does not leak
  // into the next NodeSelect.
  bool snap_overflow = false;   // the SNAPSHOT lists such an id: it stays refused until the next SetClusterSnapshot
  std::string snap_error;       // why the last snapshot was refused (kept for the NodeSelect calls that follow it)
  bool run_overflow = false;    // pack_running: some running job of the vector just packed (cached records included) holds one
  size_t mirror_ovf = 0;        // mirrored allocation records that hold one
  size_t unsupported_nodes = 0; // nodes of the snapshot flagged cns_node_soa::unsupported
  std::vector<const PdJobInScheduler*> refused;   // jobs of the last cycle the engine refused (CNS_REASON_ENGINE_REFUSED), in the cycle's order
  bool packed_from_mirror = false;
  static bool core_masks(const std::set<uint32_t>& ids, uint64_t& lo, uint64_t& hi, uint64_t& w2, uint64_t& w3) {   // true: an id >= 256
    lo = hi = w2 = w3 = 0;
    bool ovf = false;
    for (uint32_t c : ids) {
      if (c < 64) lo |= 1ull << c;
      else if (c < 128) hi |= 1ull << (c - 64);
      else if (c < 192) w2 |= 1ull << (c - 128);
      else if (c < 256) w3 |= 1ull << (c - 192);
      else ovf = true;
    }
    return ovf;
  }
  // fills `r` (a fresh or cleared ResourceInNodeV3) from the mask form; ids and slot paths arrive in ascending order,
  // so every set insertion is hinted at end()
  void fill_res(ResourceInNodeV3& r, int64_t cpu_raw, uint64_t mem, uint64_t lo, uint64_t hi, uint64_t g, uint64_t w2 = 0, uint64_t w3 = 0) const {
    r.cpu_set.cpu_count = cpu_t::from_raw(cpu_raw);
    auto& ids = r.cpu_set.core_ids;
    for (uint64_t m = lo; m; m &= m - 1) ids.insert(ids.end(), (uint32_t)__builtin_ctzll(m));
    for (uint64_t m = hi; m; m &= m - 1) ids.insert(ids.end(), 64u + (uint32_t)__builtin_ctzll(m));
    for (uint64_t m = w2; m; m &= m - 1) ids.insert(ids.end(), 128u + (uint32_t)__builtin_ctzll(m));
    for (uint64_t m = w3; m; m &= m - 1) ids.insert(ids.end(), 192u + (uint32_t)__builtin_ctzll(m));
    r.memory_bytes = mem;
    r.memory_sw_bytes = mem;
    if (g)
      for (size_t c = 0; c < classes.size(); ++c) {
        const uint64_t w = layout.class_width[c] >= 64 ? ~0ull : ((1ull << layout.class_width[c]) - 1ull);
        uint64_t bits = (g >> layout.class_shift[c]) & w;
        if (!bits) continue;
        auto& slots = r.gres[classes[c].first][classes[c].second];
        for (; bits; bits &= bits - 1) slots.insert(slots.end(), class_bit_slot[c][(uint32_t)__builtin_ctzll(bits)]);
      }
  }
  ResourceInNodeV3 to_res(int64_t cpu_raw, uint64_t mem, uint64_t lo, uint64_t hi, uint64_t g, uint64_t w2 = 0, uint64_t w3 = 0) const {
    ResourceInNodeV3 r;
    fill_res(r, cpu_raw, mem, lo, hi, g, w2, w3);
    return r;
  }

  // ---- the last cycle's packed placements, kept for the wire emission ---------------------------------------------
  struct PlacementStore {
    PinVec<int64_t> start, cpu;
    PinVec<uint8_t> reason;
    std::vector<uint8_t> excl;
    PinVec<uint64_t> off, mem, lo, hi, g, w2, w3;
    std::vector<uint64_t> msw_node, msw_task;   // msw_*: the job's memory_sw request (node + per task)
    PinVec<uint32_t> node, nt;
    size_t jobs = 0;
    explicit PlacementStore(PinCtx* c = nullptr)
        : start(PinAlloc<int64_t>(c)), cpu(PinAlloc<int64_t>(c)), reason(PinAlloc<uint8_t>(c)), off(PinAlloc<uint64_t>(c)), mem(PinAlloc<uint64_t>(c)),
          lo(PinAlloc<uint64_t>(c)), hi(PinAlloc<uint64_t>(c)), g(PinAlloc<uint64_t>(c)), w2(PinAlloc<uint64_t>(c)), w3(PinAlloc<uint64_t>(c)),
          node(PinAlloc<uint32_t>(c)), nt(PinAlloc<uint32_t>(c)) {}
  } last{&pin};
  uint64_t mem_sw_of(size_t j, uint64_t q) const {   // what write_back puts into memory_sw_bytes
    return last.excl[j] ? node_mem_sw[last.node[q]] : last.msw_node[j] + last.msw_task[j] * last.nt[q];
  }

  // ---- protobuf wire format (varint / length-delimited / fixed64), fields in number order ---------------------------
  static void put_varint(std::string& s, uint64_t v) {
    while (v >= 0x80) { s.push_back((char)(v | 0x80)); v >>= 7; }
    s.push_back((char)v);
  }
  static size_t varint_size(uint64_t v) { size_t n = 1; while (v >= 0x80) { v >>= 7; ++n; } return n; }
  static void put_bytes(std::string& s, uint32_t field, const char* p, size_t n) {
    put_varint(s, ((uint64_t)field << 3) | 2u);
    put_varint(s, n);
    s.append(p, n);
  }
  static void put_str(std::string& s, uint32_t field, const std::string& v) { put_bytes(s, field, v.data(), v.size()); }
  static void put_u64(std::string& s, uint32_t field, uint64_t v) {   // proto3: a zero scalar is not written
    if (!v) return;
    put_varint(s, (uint64_t)field << 3);
    put_varint(s, v);
  }
  // crane.grpc.DedicatedResourceInNode of a slot mask: map<name, DeviceTypeSlotsMap{map<type, Slots{repeated string}>}>
  // (PublicDefs.proto:33-44).  `classes` is sorted by (name, type), so the classes of one name are adjacent.
  void append_gres_wire(std::string& out, uint64_t g, std::string& tmp_name, std::string& tmp_type, std::string& tmp_slots) const {
    size_t c = 0;
    while (c < classes.size()) {
      size_t e = c;
      tmp_name.clear();   // DeviceTypeSlotsMap of this name
      for (; e < classes.size() && classes[e].first == classes[c].first; ++e) {
        const uint64_t w = layout.class_width[e] >= 64 ? ~0ull : ((1ull << layout.class_width[e]) - 1ull);
        uint64_t bits = (g >> layout.class_shift[e]) & w;
        if (!bits) continue;
        tmp_slots.clear();  // Slots
        for (; bits; bits &= bits - 1) put_str(tmp_slots, 1, class_bit_slot[e][(uint32_t)__builtin_ctzll(bits)]);
        tmp_type.clear();   // map entry {1: type, 2: Slots}
        put_str(tmp_type, 1, classes[e].second);
        put_str(tmp_type, 2, tmp_slots);
        put_str(tmp_name, 1, tmp_type);
      }
      if (!tmp_name.empty()) {
        tmp_type.clear();   // map entry {1: name, 2: DeviceTypeSlotsMap}
        put_str(tmp_type, 1, classes[c].first);
        put_str(tmp_type, 2, tmp_name);
        put_str(out, 1, tmp_type);
      }
      c = e;
    }
  }
  // crane.grpc.ResourceInNodeV3 (PublicDefs.proto:63-69) of one packed allocation
  void append_res_wire(std::string& out, int64_t cpu_raw, uint64_t mem, uint64_t mem_sw, uint64_t lo, uint64_t hi, uint64_t g,
                       std::string& t1, std::string& t2, std::string& t3, std::string& t4, uint64_t w2 = 0, uint64_t w3 = 0) const {
    if (lo | hi | w2 | w3) {   // repeated uint32 cpu_ids = 1, packed, ascending (std::set order)
      size_t n = 0;
      for (uint64_t m = lo; m; m &= m - 1) n += 1;                                       // ids 0..63: one byte each
      for (uint64_t m = hi; m; m &= m - 1) n += varint_size(64u + (uint32_t)__builtin_ctzll(m));
      n += 2 * (size_t)(__builtin_popcountll(w2) + __builtin_popcountll(w3));            // ids 128..255: two bytes each
      out.push_back((char)0x0A);
      put_varint(out, n);
      for (uint64_t m = lo; m; m &= m - 1) out.push_back((char)__builtin_ctzll(m));
      for (uint64_t m = hi; m; m &= m - 1) put_varint(out, 64u + (uint32_t)__builtin_ctzll(m));
      for (uint64_t m = w2; m; m &= m - 1) put_varint(out, 128u + (uint32_t)__builtin_ctzll(m));
      for (uint64_t m = w3; m; m &= m - 1) put_varint(out, 192u + (uint32_t)__builtin_ctzll(m));
    }
    if (cpu_raw != 0) {   // double cpu_count = 2: static_cast<double>(cpu_t) = raw / 2^8, exact
      const double d = (double)cpu_raw / 256.0;
      uint64_t b;
      std::memcpy(&b, &d, 8);
      out.push_back((char)0x11);
      for (int i = 0; i < 8; ++i) out.push_back((char)(b >> (8 * i)));
    }
    put_u64(out, 3, mem);
    put_u64(out, 4, mem_sw);
    t4.clear();           // DedicatedResourceInNode gres = 5: always present (mutable_gres(), PublicHeader.cpp:994)
    if (g) append_gres_wire(t4, g, t1, t2, t3);
    put_str(out, 5, t4);
  }
};

GpuNodeSelectionAlgo::GpuNodeSelectionAlgo(int device, uint64_t scheduled_batch_size) : impl_(new Impl) {
  cns_config cfg{};
  cfg.abi_version = CNS_ABI_VERSION;
  cfg.device = device;
  cfg.scheduled_batch_size = scheduled_batch_size;
  batch_ = scheduled_batch_size;
  status_ = cns_create(&cfg, &impl_->h);
  if (status_ != 0) error_ = cns_last_error(nullptr);
  impl_->pin.h = impl_->h;   // (null without an engine: the cycle's arrays are then plain memory)
}

// Several devices of one node (BASELINE.json: "job-sharded across 8 x MI355X"): ONE algorithm object, as in the reference
// (JobScheduler.cpp:158-159), over one engine per device; the groups of partitions connected through shared nodes are dealt round robin
// (their LocalSchedulers are independent: :6723-6732,6746-6761), every device runs its shard on its own host thread, the packed results
// are all-gathered on the devices and merged back into the PdJobInSchedulers in queue order (cns_group_select).  The cycle with preemption
// and the kernels either side of the path (run limits, steps) stay on the first device.
GpuNodeSelectionAlgo::GpuNodeSelectionAlgo(const std::vector<int>& devices, uint64_t scheduled_batch_size) : impl_(new Impl) {
  cns_config cfg{};
  cfg.abi_version = CNS_ABI_VERSION;
  cfg.device = devices.empty() ? 0 : devices[0];
  cfg.scheduled_batch_size = scheduled_batch_size;
  batch_ = scheduled_batch_size;
  impl_->devices = devices.empty() ? std::vector<int>{0} : devices;
  if (impl_->devices.size() == 1) {
    status_ = cns_create(&cfg, &impl_->h);
    if (status_ != 0) error_ = cns_last_error(nullptr);
  } else {
    std::vector<int32_t> dev(impl_->devices.begin(), impl_->devices.end());
    status_ = cns_group_create(&cfg, dev.data(), (uint32_t)dev.size(), &impl_->grp);
    if (status_ != 0) error_ = cns_group_last_error(nullptr);
    else impl_->h = cns_group_handle(impl_->grp, 0);
  }
  impl_->pin.h = impl_->h;
}

size_t GpuNodeSelectionAlgo::NumDevices() const { return impl_->devices.empty() ? 1 : impl_->devices.size(); }

bool GpuNodeSelectionAlgo::LastGroupInfo(cns_group_info* out) const {
  return impl_->grp && out && cns_group_get_info(impl_->grp, out) == 0;
}

void GpuNodeSelectionAlgo::SetCranedState(const CranedId& craned_id, bool alive, bool drain) {
  Impl& I = *impl_;
  auto it = I.node_idx.find(craned_id);
  if (!I.have_snapshot || it == I.node_idx.end()) { status_ = CNS_ERR_INVALID_ARG; error_ = "SetCranedState: unknown craned or no snapshot"; return; }
  const uint8_t s = alive && !drain;   // JobScheduler.cpp:6595
  if (I.n_sched[it->second] == s) return;
  I.n_sched[it->second] = s;
  status_ = I.push_tables(error_);     // the packed tables again, no string is looked at; dense indices stay valid
  if (status_ == 0) error_.clear();
}

void GpuNodeSelectionAlgo::PendingCycleForBench(const std::vector<std::unique_ptr<PdJobInScheduler>>& pending_jobs,
                                                double* pack_ms, double* write_back_ms, uint64_t* checksum) {
  Impl& I = *impl_;
  std::vector<PdJobInScheduler*> ord;
  for (const auto& j : pending_jobs) ord.push_back(j.get());
  Impl::PackedJobs B;
  auto t0 = std::chrono::steady_clock::now();
  I.pack_pending(ord, B);
  *pack_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  // synthetic placements: job j starts now on node j % N with the 4 lowest cores (what cns_select would hand back)
  const size_t J = ord.size(), N = std::max<size_t>(I.node_name.size(), 1);
  std::vector<int64_t> st(J, 1000), cpu(J, 4 * 256);
  std::vector<uint8_t> rs(J, 0);
  std::vector<uint64_t> off(J + 1), mem(J, 1ull << 30), lo(J, 0xF), hi(J, 0), g(J, 0);
  std::vector<uint32_t> node(J), ntk(J, 1);
  for (size_t j = 0; j < J; ++j) { off[j] = j; node[j] = (uint32_t)(j % N); }
  off[J] = J;
  cns_placement_soa o{};
  o.place_capacity = J; o.start_sec = st.data(); o.reason = rs.data(); o.place_offsets = off.data(); o.node_idx = node.data();
  o.ntasks = ntk.data(); o.cpu_raw = cpu.data(); o.mem = mem.data(); o.core_lo = lo.data(); o.core_hi = hi.data(); o.gres = g.data();
  t0 = std::chrono::steady_clock::now();
  I.write_back(ord, o);
  *write_back_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  {  // the same placements as "the last cycle" of the wire emission (EmitWireForBench)
    Impl::PlacementStore& S = I.last;
    S.start.assign(st.begin(), st.end()); S.cpu.assign(cpu.begin(), cpu.end()); S.reason.assign(rs.begin(), rs.end()); S.off.assign(off.begin(), off.end());
    S.mem.assign(mem.begin(), mem.end()); S.lo.assign(lo.begin(), lo.end()); S.hi.assign(hi.begin(), hi.end()); S.g.assign(g.begin(), g.end());
    S.node.assign(node.begin(), node.end()); S.nt.assign(ntk.begin(), ntk.end()); S.w2.assign(J, 0); S.w3.assign(J, 0);
    S.excl.assign(J, 0); S.msw_node.assign(J, 0); S.msw_task.assign(J, 0);
    S.jobs = J;
    I.last_ord.assign(ord.begin(), ord.end());
    I.last_index.clear();
  }
  uint64_t hsh = 1469598103934665603ull;
  auto mix = [&](const void* p, size_t n) { const unsigned char* c = (const unsigned char*)p; for (size_t i = 0; i < n; ++i) { hsh ^= c[i]; hsh *= 1099511628211ull; } };
  mix(B.part.data(), J * 4); mix(B.L.data(), J * 8); mix(B.tcpu.data(), J * 8); mix(B.tmem.data(), J * 8); mix(B.k.data(), J * 4);
  mix(B.gtot.data(), B.gtot.size()); mix(B.gspec.data(), B.gspec.size()); mix(B.skip.data(), J);
  for (size_t j = 0; j < J; j += 997) { const auto& p = *ord[j]; mix(p.craned_ids[0].data(), p.craned_ids[0].size()); mix(&p.end_time, 8); }
  *checksum = hsh;
}

size_t GpuNodeSelectionAlgo::PackRunningForBench(const std::vector<std::unique_ptr<RnJobInScheduler>>& running_jobs,
                                                 bool use_cache, uint64_t* checksum, double* pack_ms) {
  Impl& I = *impl_;
  I.use_alloc_cache = use_cache;
  const auto t0 = std::chrono::steady_clock::now();
  I.pack_running(running_jobs);
  if (pack_ms) *pack_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  I.use_alloc_cache = true;
  if (checksum) {  // FNV-1a over everything cns_set_running would receive
    uint64_t hsh = 1469598103934665603ull;
    auto mix = [&](const void* p, size_t n) { const unsigned char* c = (const unsigned char*)p; for (size_t i = 0; i < n; ++i) { hsh ^= c[i]; hsh *= 1099511628211ull; } };
    mix(I.r_end.data(), I.r_end.size() * 8); mix(I.r_off.data(), I.r_off.size() * 4); mix(I.r_node.data(), I.r_node.size() * 4);
    mix(I.r_cpu.data(), I.r_cpu.size() * 8); mix(I.r_mem.data(), I.r_mem.size() * 8); mix(I.r_lo.data(), I.r_lo.size() * 8);
    mix(I.r_hi.data(), I.r_hi.size() * 8); mix(I.r_g.data(), I.r_g.size() * 8); mix(I.r_resv.data(), I.r_resv.size() * 4);
    // (the planes of core ids 128..255 only when any is set: the checksums of runs without such nodes stay what they were)
    bool anyw = false;
    for (size_t i = 0; i < I.r_w2.size(); ++i) anyw = anyw || (I.r_w2[i] | I.r_w3[i]) != 0;
    if (anyw) { mix(I.r_w2.data(), I.r_w2.size() * 8); mix(I.r_w3.data(), I.r_w3.size() * 8); }
    *checksum = hsh;
  }
  return I.r_node.size();
}

GpuNodeSelectionAlgo::~GpuNodeSelectionAlgo() {
  if (impl_) {   // the page-locked arrays go back first: cns_destroy releases whatever the handle still owns
    impl_->packed = Impl::PackedJobs(nullptr);
    impl_->last = Impl::PlacementStore(nullptr);
    impl_->pin.h = nullptr;
  }
  if (impl_ && impl_->grp) cns_group_destroy(impl_->grp);   // (its engines, the first one — impl_->h — included)
  else if (impl_ && impl_->h) cns_destroy(impl_->h);
}

void GpuNodeSelectionAlgo::LastCycleMs(double* pack_ms, double* engine_ms, double* write_back_ms) const {
  if (pack_ms) *pack_ms = impl_->t_pack_ms;
  if (engine_ms) *engine_ms = impl_->t_engine_ms;
  if (write_back_ms) *write_back_ms = impl_->t_write_ms;
}

void GpuNodeSelectionAlgo::SetFullWriteBack(bool full) { impl_->lazy_write_back = !full; }
void GpuNodeSelectionAlgo::SetDeferredWriteBack(bool deferred) { impl_->deferred_write_back = deferred; }
void GpuNodeSelectionAlgo::SetHostThreads(int n) {
  impl_->host_threads = n < 1 ? 1 : n;
  const uint32_t e = (uint32_t)std::min(impl_->host_threads, 64);
  if (impl_->grp) { for (uint32_t d = 0; d < cns_group_size(impl_->grp); ++d) (void)cns_set_host_threads(cns_group_handle(impl_->grp, d), e); }
  else if (impl_->h) (void)cns_set_host_threads(impl_->h, e);
}

bool GpuNodeSelectionAlgo::MaterializeAllocation(PdJobInScheduler& job) {
  Impl& I = *impl_;
  const Impl::PlacementStore& S = I.last;
  if (I.last_index.size() != I.last_ord.size()) {
    I.last_index.clear();
    I.last_index.reserve(I.last_ord.size());
    for (size_t j = 0; j < I.last_ord.size(); ++j) I.last_index[I.last_ord[j]] = j;
  }
  auto li = I.last_index.find(&job);
  if (li == I.last_index.end() || li->second >= S.jobs || S.start[li->second] == 0) return false;
  const size_t j = li->second;
  job.craned_id_to_task_num.clear();
  job.allocated_res.clear();
  for (uint64_t q = S.off[j]; q < S.off[j + 1]; ++q) {
    if (S.node[q] == CNS_NODE_NONE) continue;
    const CranedId& cid = I.node_name[S.node[q]];
    job.craned_id_to_task_num[cid] = S.nt[q];
    ResourceInNodeV3& res = job.allocated_res[cid];
    I.fill_res(res, S.cpu[q], S.mem[q], S.lo[q], S.hi[q], S.g[q], S.w2[q], S.w3[q]);
    res.memory_sw_bytes = I.mem_sw_of(j, q);
  }
  return true;
}

void GpuNodeSelectionAlgo::SetClusterSnapshot(const ClusterSnapshot& snap) {
  Impl& I = *impl_;
  I.have_snapshot = false;
  I.alloc_cache.clear();   // dense node indices and GRES bit positions are per snapshot
  const uint32_t N = (uint32_t)snap.craned_metas.size();
  I.node_name.clear(); I.node_mem_sw.clear(); I.node_idx.clear(); I.part_idx.clear();
  bool core_overflow = false;
  I.preempt_enabled = snap.preempt_enabled;
  I.qos_id.clear(); I.qos_preempt.clear();
  if (snap.preempt_enabled) {
    for (const auto& [name, lst] : snap.qos_preempt) I.qos_of(name);
    for (const auto& [name, lst] : snap.qos_preempt) {
      std::vector<uint32_t> ids;
      for (const auto& p : lst) ids.push_back(I.qos_of(p));
      I.qos_preempt[I.qos_id.at(name)] = ids;
    }
  }
  I.classes.clear(); I.name_id.clear(); I.class_slot_bit.clear(); I.class_bit_slot.clear();
  // GRES classes: every (name,type) seen in any res_total; bits per class = union of its slot paths
  std::map<std::pair<std::string, std::string>, std::set<SlotId>> cls;
  for (const auto& m : snap.craned_metas)
    for (const auto& [name, tm] : m.res_total.gres)
      for (const auto& [type, slots] : tm) cls[{name, type}].insert(slots.begin(), slots.end());
  memset(&I.layout, 0, sizeof I.layout);
  uint32_t shift = 0;
  std::set<std::pair<std::string, std::string>> cls_out;   // (name, type) classes the 64-bit slot mask has no room for: the nodes that carry them are
                                                            // flagged unsupported below — their partitions go to the CPU scheduler, the others stay here
  for (const auto& [key, slots] : cls) {
    if (I.classes.size() >= CNS_MAX_GRES_CLASSES || shift + slots.size() > 64 ||
        (!I.name_id.count(key.first) && I.name_id.size() >= CNS_MAX_GRES_NAMES)) { cls_out.insert(key); continue; }
    if (!I.name_id.count(key.first)) {
      uint32_t id = (uint32_t)I.name_id.size();
      I.name_id[key.first] = id;
    }
    const uint32_t c = (uint32_t)I.classes.size();
    I.classes.push_back(key);
    I.layout.class_name[c] = (uint8_t)I.name_id[key.first];
    I.layout.class_shift[c] = (uint8_t)shift;
    I.layout.class_width[c] = (uint8_t)slots.size();
    I.class_slot_bit.emplace_back();
    I.class_bit_slot.emplace_back();
    for (const auto& s : slots) {  // std::set order = lexicographic path order
      I.class_slot_bit[c][s] = shift + (uint32_t)I.class_bit_slot[c].size();
      I.class_bit_slot[c].push_back(s);
    }
    shift += (uint32_t)slots.size();
  }
  I.layout.num_classes = (uint32_t)I.classes.size();

  auto &cpu = I.n_cpu; auto &mem = I.n_mem, &lo = I.n_lo, &hi = I.n_hi, &gres = I.n_gres, &w2 = I.n_w2, &w3 = I.n_w3;
  auto &sched = I.n_sched;
  cpu.assign(N, 0); mem.assign(N, 0); lo.assign(N, 0); hi.assign(N, 0); gres.assign(N, 0); sched.assign(N, 0);
  w2.assign(N, 0); w3.assign(N, 0);
  I.n_unsup.assign(N, 0);
  size_t n_unsupported = 0;
  for (uint32_t n = 0; n < N; ++n) {
    const CranedMeta& m = snap.craned_metas[n];
    I.node_name.push_back(m.craned_id);
    I.node_mem_sw.push_back(m.res_total.memory_sw_bytes);
    I.node_idx[m.craned_id] = n;
    cpu[n] = m.res_total.cpu_set.cpu_count.raw;
    mem[n] = m.res_total.memory_bytes;
    bool out = I.core_masks(m.res_total.cpu_set.core_ids, lo[n], hi[n], w2[n], w3[n]);   // a core id >= 256
    for (const auto& [name, tm] : m.res_total.gres)
      for (const auto& [type, slots] : tm) out = out || (!slots.empty() && cls_out.count({name, type}) != 0);
    gres[n] = I.gres_mask(m.res_total.gres);
    sched[n] = m.alive && !m.drain;  // JobScheduler.cpp:6595
    if (out) { I.n_unsup[n] = 1; ++n_unsupported; }
  }
  I.unsupported_nodes = n_unsupported;
  auto &poff = I.n_poff, &pnodes = I.n_pnodes;
  poff.assign(1, 0); pnodes.clear();
  for (const auto& [pid, ids] : snap.partitions) {
    I.part_idx[pid] = (uint32_t)poff.size() - 1;
    for (const auto& id : ids) {
      auto it = I.node_idx.find(id);
      if (it != I.node_idx.end()) pnodes.push_back(it->second);
    }
    poff.push_back((uint32_t)pnodes.size());
  }
  // ---- reservations (JobScheduler.cpp:6619-6679) ----
  I.resv_idx.clear();
  const uint32_t V = (uint32_t)snap.reservations.size();
  I.v_start.assign(V, 0); I.v_end.assign(V, 0);
  I.v_off.assign(1, 0); I.v_node.clear(); I.v_cpu.clear(); I.v_mem.clear(); I.v_lo.clear(); I.v_hi.clear(); I.v_g.clear(); I.v_w2.clear(); I.v_w3.clear();
  for (uint32_t v = 0; v < V; ++v) {
    const ResvMeta& m = snap.reservations[v];
    I.resv_idx[m.name] = v;
    I.v_start[v] = m.start_time; I.v_end[v] = m.end_time;
    for (const auto& [cid, res] : m.res_total) {
      auto it = I.node_idx.find(cid);
      if (it == I.node_idx.end()) continue;
      I.v_node.push_back(it->second);
      I.v_cpu.push_back(res.cpu_set.cpu_count.raw);
      I.v_mem.push_back(res.memory_bytes);
      uint64_t l, hh, x2, x3;
      core_overflow |= I.core_masks(res.cpu_set.core_ids, l, hh, x2, x3);
      I.v_lo.push_back(l); I.v_hi.push_back(hh); I.v_w2.push_back(x2); I.v_w3.push_back(x3);
      I.v_g.push_back(I.gres_mask(res.gres));
    }
    I.v_off.push_back((uint32_t)I.v_node.size());
  }
  I.repack_mirror();
  I.snap_overflow = core_overflow;
  I.snap_error.clear();
  if (core_overflow) {
    status_ = CNS_ERR_UNSUPPORTED;
    error_ = I.snap_error = "a reservation's share lists a core id >= 256 (the engine keeps core ids in four 64-bit masks); keep the CPU SchedulerAlgo";
    return;
  }
  if (!I.h) return;   // no device: the dictionaries above still serve PackRunningForBench
  status_ = I.push_tables(error_);
  if (status_ != 0) return;
  I.have_snapshot = true;
}

// ---- event-fed mirror (the calls CranedMetaContainer gets, CranedMetaContainer.cpp:178-277) ------------------------------
void GpuNodeSelectionAlgo::MallocResourceFromNode(const CranedId& craned_id, job_id_t job_id, const ResourceV3& resources) {
  Impl& I = *impl_;
  auto rit = resources.find(craned_id);   // resources.At(node_id), :198
  if (rit == resources.end()) return;
  auto& mj = I.mirror[job_id];
  if (I.mirror_packed && job_id <= I.m_last_id) I.m_touched.push_back(job_id);   // a job inside the packed range changes (or appears there)
  Impl::MirrorRec* rec = nullptr;
  for (auto& r : mj.recs) if (r.craned == craned_id) rec = &r;   // (rn_job_res_map.emplace: one record per (craned, job))
  if (!rec) { mj.recs.emplace_back(); rec = &mj.recs.back(); rec->craned = craned_id; }
  rec->res = rit->second;
  I.pack_rec(*rec);
}

void GpuNodeSelectionAlgo::FreeResourceFromNode(const CranedId& craned_id, job_id_t job_id) {
  Impl& I = *impl_;
  auto it = I.mirror.find(job_id);
  if (it == I.mirror.end()) return;   // "Try to free resource from an unknown job", :247-251
  auto& recs = it->second.recs;
  for (size_t i = 0; i < recs.size(); ++i)
    if (recs[i].craned == craned_id) {
      if (recs[i].packed.ovf) --I.mirror_ovf;
      recs.erase(recs.begin() + i);
      break;
    }
  if (I.mirror_packed && job_id <= I.m_last_id) {
    if (recs.empty()) I.m_removed.push_back(job_id);
    else I.m_touched.push_back(job_id);   // (the rest of its nodes usually follow before the next cycle: then it is a removal)
  }
  if (recs.empty()) I.mirror.erase(it);
}

void GpuNodeSelectionAlgo::SetRunningJobInfo(job_id_t job_id, TimeSec end_time, const std::string& reservation) {
  auto& mj = impl_->mirror[job_id];
  mj.end_time = end_time;
  mj.resv = reservation;
  if (impl_->mirror_packed && job_id <= impl_->m_last_id) impl_->m_info.push_back(job_id);
}

size_t GpuNodeSelectionAlgo::MirroredRunningJobs() const { return impl_->mirror.size(); }
void GpuNodeSelectionAlgo::MirrorPackCounts(size_t* full_walks, size_t* patches) const {
  if (full_walks) *full_walks = impl_->mirror_full_packs;
  if (patches) *patches = impl_->mirror_patch_packs;
}

size_t GpuNodeSelectionAlgo::PackMirrorForBench(uint64_t* checksum_canonical, double* pack_ms) {
  Impl& I = *impl_;
  const auto t0 = std::chrono::steady_clock::now();
  I.pack_from_mirror();
  if (pack_ms) *pack_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (checksum_canonical) *checksum_canonical = I.running_checksum_canonical();
  return I.r_node.size();
}

uint64_t GpuNodeSelectionAlgo::LastRunningChecksumCanonical() const { return impl_->running_checksum_canonical(); }
bool GpuNodeSelectionAlgo::PackedRunningSetOverflows() const { return impl_->packed_from_mirror ? impl_->mirror_ovf != 0 : impl_->run_overflow; }

void GpuNodeSelectionAlgo::NodeSelect(const TimeSec& now, const std::vector<std::unique_ptr<PdJobInScheduler>>& pending_jobs,
                                      const std::vector<std::unique_ptr<RnJobInScheduler>>* running_for_priority) {
  static const std::vector<std::unique_ptr<RnJobInScheduler>> kNone;
  impl_->pack_from_mirror();
  SelectPacked_(now, pending_jobs, running_for_priority ? *running_for_priority : kNone);
}

void GpuNodeSelectionAlgo::NodeSelect(const TimeSec& now,
                                      const std::vector<std::unique_ptr<RnJobInScheduler>>& running_jobs,
                                      const std::vector<std::unique_ptr<PdJobInScheduler>>& pending_jobs) {
  // ---- running jobs (JobScheduler.cpp:6681-6709), packed incrementally -----------------------------------------
  if (impl_->h && impl_->have_snapshot) impl_->pack_running(running_jobs);
  SelectPacked_(now, pending_jobs, running_jobs);
}

// the cycle proper, on the running allocations already packed in Impl::r_*
// LicenseManager::CheckLicenseCountSufficient (LicenseManager.cpp:167-221), the pre-pass NodeSelect runs between ordering and selection
// (JobScheduler.cpp:6739): a sequential counter pass over the ordered jobs on the cycle's working copy of the license table.  Host-only
// and static, so that it is pinned to the reference's own compiled function without a device (tests/test_ref_pin.py).
void GpuNodeSelectionAlgo::CheckLicenseCountSufficient(const std::unordered_map<std::string, License>& licenses, const std::vector<PdJobInScheduler*>& ord) {
  std::unordered_map<std::string, License> avail = licenses;  // the cycle's working copy (:169-176)
  for (PdJobInScheduler* job : ord) {
    if (job->req_licenses.empty()) continue;
    job->actual_licenses.clear();
    if (job->is_license_or) {  // first alternative that fits (:183-194)
      for (const auto& [key, count] : job->req_licenses) {
        auto it = avail.find(key);
        if (it == avail.end()) continue;
        const License& lic = it->second;
        if ((uint32_t)(count + lic.reserved + lic.used + lic.last_deficit) <= lic.total) {   // (uint32 arithmetic, as the reference's: :188-189)
          job->actual_licenses.emplace(key, count);
          break;
        }
      }
    } else {                   // all of them (:195-210)
      for (const auto& [key, count] : job->req_licenses) {
        auto it = avail.find(key);
        if (it == avail.end()) { job->actual_licenses.clear(); break; }
        const License& lic = it->second;
        if ((uint32_t)(count + lic.reserved + lic.used + lic.last_deficit) > lic.total) { job->actual_licenses.clear(); break; }
        job->actual_licenses.emplace(key, count);
      }
    }
    if (job->actual_licenses.empty()) { job->reason = "License"; continue; }  // :212-215
    for (const auto& [key, count] : job->actual_licenses) avail[key].used += count;  // :217-219
  }
}

void GpuNodeSelectionAlgo::SelectPacked_(const TimeSec& now, const std::vector<std::unique_ptr<PdJobInScheduler>>& pending_jobs,
                                         const std::vector<std::unique_ptr<RnJobInScheduler>>& running_jobs) {
  Impl& I = *impl_;
  auto fail_all = [&](int st, const std::string& msg) {
    status_ = st; error_ = msg;
    for (const auto& j : pending_jobs) if (j->reason.empty()) j->reason = "GpuEngineError";
    I.last.jobs = 0; I.last_ord.clear(); I.last_index.clear();   // MaterializeAllocation must not serve the PREVIOUS cycle's placements
  };
  if (!I.h) return fail_all(status_ ? status_ : CNS_ERR_NO_DEVICE, error_);
  if (!I.have_snapshot) {
    if (I.snap_overflow) return fail_all(CNS_ERR_UNSUPPORTED, I.snap_error);   // the refused snapshot's own message, every cycle
    return fail_all(CNS_ERR_STATE, "NodeSelect before SetClusterSnapshot");
  }
  if (I.packed_from_mirror ? I.mirror_ovf != 0 : I.run_overflow)   // (judged on the records of THIS cycle's running set, cached or fresh)
    return fail_all(CNS_ERR_UNSUPPORTED, "a running job holds a core id >= 256");
  const auto &r_end = I.r_end, &r_cpu = I.r_cpu;
  const auto &r_off = I.r_off, &r_node = I.r_node, &r_resv = I.r_resv;
  const auto &r_mem = I.r_mem, &r_lo = I.r_lo, &r_hi = I.r_hi, &r_g = I.r_g;
  cns_running_soa rs{};
  rs.num_jobs = (uint32_t)r_end.size(); rs.num_allocs = (uint32_t)r_node.size();
  rs.end_sec = r_end.data(); rs.alloc_offsets = r_off.data(); rs.alloc_node = r_node.data();
  rs.alloc_cpu_raw = r_cpu.data(); rs.alloc_mem = r_mem.data(); rs.alloc_core_lo = r_lo.data();
  rs.alloc_core_hi = r_hi.data(); rs.alloc_gres = r_g.data(); rs.reservation = r_resv.data();
  rs.alloc_core_w2 = I.r_w2.data(); rs.alloc_core_w3 = I.r_w3.data();
  int st = I.grp ? cns_group_set_running(I.grp, rs.num_jobs ? &rs : nullptr) : cns_set_running(I.h, rs.num_jobs ? &rs : nullptr);
  if (st != 0) return fail_all(st, I.grp ? cns_group_last_error(I.grp) : cns_last_error(I.h));

  // ---- pending jobs, in the sorter's order (JobScheduler.cpp:6735) ---------------------------------------
  std::vector<PdJobInScheduler*> ord;
  if (sorter_) {
    sorter_->GetOrderedJobPtrVec(now, pending_jobs, running_jobs, batch_ ? (size_t)batch_ : pending_jobs.size(), ord);
  } else {
    ord.reserve(pending_jobs.size());
    for (const auto& j : pending_jobs) ord.push_back(j.get());
  }
  // ---- license pre-pass (JobScheduler.cpp:6739) ---------------------------------------------------------------------------------
  CheckLicenseCountSufficient(licenses_, ord);
  const size_t J = ord.size();
  const auto tp0 = std::chrono::steady_clock::now();
  Impl::PackedJobs& B = I.packed;
  Impl::PlacementStore& S = I.last;   // kept after the cycle: the wire emission reads the packed placements
  S.jobs = 0;
  uint64_t places = 0;
  I.pack_pending(ord, B, &S, &places);
  auto &part = B.part, &k = B.k, &nt = B.nt, &tmin = B.tmin, &tmax = B.tmax, &inodes = B.inodes, &enodes = B.enodes, &jresv = B.jresv;
  auto &L = B.L, &ncpu = B.ncpu, &tcpu = B.tcpu;
  auto &nmem = B.nmem, &tmem = B.tmem, &ioff = B.ioff, &eoff = B.eoff;
  auto &excl = B.excl, &skip = B.skip, &gtot = B.gtot, &gspec = B.gspec;
  cns_job_soa js{};
  js.num_jobs = J;
  js.partition = part.data(); js.time_limit_sec = L.data(); js.node_cpu_raw = ncpu.data(); js.node_mem = nmem.data();
  js.task_cpu_raw = tcpu.data(); js.task_mem = tmem.data(); js.node_num = k.data(); js.ntasks = nt.data();
  js.ntasks_per_node_min = tmin.data(); js.ntasks_per_node_max = tmax.data(); js.exclusive = excl.data();
  js.gres_total = gtot.data(); js.gres_spec = gspec.data(); js.incl_offsets = ioff.data(); js.incl_nodes = inodes.data();
  js.excl_offsets = eoff.data(); js.excl_nodes = enodes.data(); js.skip = skip.data(); js.reservation = jresv.data();

  // (sized, not zero-filled: cns_download / cns_group_select write every job's start and reason and every placement record of the cycle)
  S.start.resize(J + 1); S.cpu.resize(places + 1); S.reason.resize(J + 1); S.off.resize(J + 1);
  S.mem.resize(places + 1); S.lo.resize(places + 1); S.hi.resize(places + 1); S.g.resize(places + 1);
  S.w2.resize(places + 1); S.w3.resize(places + 1);
  S.node.resize(places + 1); S.nt.resize(places + 1);
  cns_placement_soa out{};
  out.place_capacity = places;
  out.start_sec = S.start.data(); out.reason = S.reason.data(); out.place_offsets = S.off.data();
  out.node_idx = S.node.data(); out.ntasks = S.nt.data(); out.cpu_raw = S.cpu.data(); out.mem = S.mem.data();
  out.core_lo = S.lo.data(); out.core_hi = S.hi.data(); out.gres = S.g.data(); out.core_w2 = S.w2.data(); out.core_w3 = S.w3.data();
  I.last_index.clear();
  I.last_ord.clear();
  I.cancelled.clear();
  const auto tp1 = std::chrono::steady_clock::now();
  I.t_pack_ms = std::chrono::duration<double, std::milli>(tp1 - tp0).count();
  if (!I.preempt_enabled) {
    st = I.grp ? cns_group_select(I.grp, now, &js, &out) : cns_select(I.h, now, &js, &out);
    if (st != 0) return fail_all(st, I.grp ? cns_group_last_error(I.grp) : cns_last_error(I.h));
  } else if (I.grp) {
    return fail_all(CNS_ERR_UNSUPPORTED, "a cycle with preemption runs on ONE device (TryPreempt_ releases resources inside the cycle: csrc/preempt_dev.inc): build the algorithm over one device for it");
  } else {
    // ---- the cycle with preemption (include/crane_gpu/preempt.h): qos ids, the fields TryPreempt_ reads, the set ------
    if (!I.r_src_valid || I.r_src.size() != I.r_end.size())
      return fail_all(CNS_ERR_STATE, "preemption needs the running jobs themselves: call NodeSelect(now, running_jobs, pending_jobs)");
    const uint32_t R = (uint32_t)I.r_src.size();
    std::vector<uint32_t> pj_id(J + 1), pj_qos(J + 1), pj_qp(J + 1), rj_id(R + 1), rj_qos(R + 1), rj_qp(R + 1), pset;
    std::vector<double> pj_prio(J + 1);
    std::vector<int64_t> rj_start(R + 1);
    for (size_t j = 0; j < J; ++j) { pj_id[j] = ord[j]->job_id; pj_qos[j] = I.qos_of(ord[j]->qos); pj_qp[j] = ord[j]->qos_priority; pj_prio[j] = ord[j]->priority; }
    for (uint32_t r = 0; r < R; ++r) { rj_id[r] = I.r_src[r]->job_id; rj_qos[r] = I.qos_of(I.r_src[r]->qos); rj_qp[r] = I.r_src[r]->qos_priority; rj_start[r] = I.r_src[r]->start_time; }
    for (job_id_t id : I.preempting) pset.push_back(id);
    std::vector<uint32_t> qoff(I.qos_preempt.size() + 1, 0), qflat;
    for (size_t q = 0; q < I.qos_preempt.size(); ++q) { for (uint32_t x : I.qos_preempt[q]) qflat.push_back(x); qoff[q + 1] = (uint32_t)qflat.size(); }
    if (qflat.empty()) qflat.push_back(0);
    if (pset.empty()) pset.push_back(0);
    cns_preempt_soa ps{};
    ps.enabled = 1; ps.num_qos = (uint32_t)I.qos_preempt.size();
    ps.qos_preempt_offsets = qoff.data(); ps.qos_preempt = qflat.data();
    ps.pd_job_id = pj_id.data(); ps.pd_qos = pj_qos.data(); ps.pd_qos_priority = pj_qp.data(); ps.pd_priority = pj_prio.data();
    ps.rn_job_id = rj_id.data(); ps.rn_qos = rj_qos.data(); ps.rn_qos_priority = rj_qp.data(); ps.rn_start_sec = rj_start.data();
    ps.num_preempting = (uint32_t)I.preempting.size(); ps.preempting_job_ids = pset.data();
    std::vector<uint64_t> po_off(J + 1, 0);
    std::vector<uint32_t> po_refs(4 * (J + R) + 64), po_cancel(R + 1), po_set(R + I.preempting.size() + 1);
    cns_preempt_out po{};
    po.capacity = po_refs.size(); po.offsets = po_off.data(); po.preempted = po_refs.data();
    po.cancel_capacity = (uint32_t)po_cancel.size(); po.cancelled_job_ids = po_cancel.data();
    po.preempting_capacity = (uint32_t)po_set.size(); po.preempting_job_ids = po_set.data();
    st = cns_select_preempt(I.h, now, &js, &ps, &out, &po);
    if (st != 0) return fail_all(st, cns_last_error(I.h));
    for (size_t j = 0; j < J; ++j)
      for (uint64_t x = po_off[j]; x < po_off[j + 1]; ++x) {
        const uint32_t ref = po_refs[x];
        if (ref & CNS_PREEMPT_REF_PENDING) ord[j]->preempted_jobs.emplace_back(ord[ref & 0x7FFFFFFFu]);
        else ord[j]->preempted_jobs.emplace_back(I.r_src[ref]);
      }
    I.cancelled.assign(po_cancel.begin(), po_cancel.begin() + po.num_cancelled);
    I.preempting.clear();
    for (uint32_t i = 0; i < po.num_preempting; ++i) I.preempting.insert(po_set[i]);
  }
  I.last_ord.assign(ord.begin(), ord.end());
  I.refused.clear();
  for (size_t j = 0; j < J; ++j) if (out.reason[j] == CNS_REASON_ENGINE_REFUSED) I.refused.push_back(ord[j]);
  S.jobs = J;
  status_ = 0;
  error_.clear();
  const auto tp2 = std::chrono::steady_clock::now();
  I.write_back(ord, out);
  const auto tp3 = std::chrono::steady_clock::now();
  I.t_engine_ms = std::chrono::duration<double, std::milli>(tp2 - tp1).count();
  I.t_write_ms = std::chrono::duration<double, std::milli>(tp3 - tp2).count();
}

// ---------------------------------------------------------------------------------------------------------
// Placement -> wire
// ---------------------------------------------------------------------------------------------------------
const std::vector<const PdJobInScheduler*>& GpuNodeSelectionAlgo::LastOrder() const { return impl_->last_ord; }
const std::vector<job_id_t>& GpuNodeSelectionAlgo::LastPreemptCancel() const { return impl_->cancelled; }
const std::vector<const PdJobInScheduler*>& GpuNodeSelectionAlgo::RefusedJobs() const { return impl_->refused; }
size_t GpuNodeSelectionAlgo::UnsupportedNodes() const { return impl_->unsupported_nodes; }
std::vector<PartitionId> GpuNodeSelectionAlgo::RefusedPartitions() const {
  std::vector<PartitionId> out;
  const Impl& I = *impl_;
  if (!I.h || !I.have_snapshot) return out;
  std::vector<uint8_t> st(I.part_idx.size() + 1, 0);
  if ((I.grp ? cns_group_get_partition_status(I.grp, st.data(), (uint32_t)st.size()) : cns_get_partition_status(I.h, st.data(), (uint32_t)st.size())) != 0) return out;
  for (const auto& [pid, idx] : I.part_idx) if (idx < st.size() && st[idx]) out.push_back(pid);
  std::sort(out.begin(), out.end());
  return out;
}
const std::set<job_id_t>& GpuNodeSelectionAlgo::PreemptingSet() const { return impl_->preempting; }

size_t GpuNodeSelectionAlgo::EmitStartedResourcesWire(WireBatch* out) const {
  const Impl& I = *impl_;
  const Impl::PlacementStore& S = I.last;
  out->bytes.clear();
  out->recs.clear();
  std::string t1, t2, t3, t4;
  for (size_t j = 0; j < S.jobs; ++j) {
    if (S.reason[j] != CNS_REASON_NONE || S.start[j] == 0) continue;   // only a job that starts now is dispatched (cpp:1507-1510)
    for (uint64_t q = S.off[j]; q < S.off[j + 1]; ++q) {
      if (S.node[q] == CNS_NODE_NONE) continue;
      const size_t at = out->bytes.size();
      I.append_res_wire(out->bytes, S.cpu[q], S.mem[q], I.mem_sw_of(j, q), S.lo[q], S.hi[q], S.g[q], t1, t2, t3, t4, S.w2[q], S.w3[q]);
      out->recs.push_back({(uint32_t)j, S.node[q], (uint32_t)at, (uint32_t)(out->bytes.size() - at)});
    }
  }
  return out->recs.size();
}

bool GpuNodeSelectionAlgo::AppendResourceInNodeV3Wire(const PdJobInScheduler& job, const CranedId& craned_id, std::string* out) {
  Impl& I = *impl_;
  const Impl::PlacementStore& S = I.last;
  if (I.last_index.size() != I.last_ord.size()) {
    I.last_index.clear();
    I.last_index.reserve(I.last_ord.size());
    for (size_t j = 0; j < I.last_ord.size(); ++j) I.last_index[I.last_ord[j]] = j;
  }
  auto li = I.last_index.find(&job);
  auto ni = I.node_idx.find(craned_id);
  if (li == I.last_index.end() || ni == I.node_idx.end() || li->second >= S.jobs || S.start[li->second] == 0 ||
      S.reason[li->second] != CNS_REASON_NONE)   // only a job that starts now is dispatched
    return false;
  const size_t j = li->second;
  std::string t1, t2, t3, t4;
  for (uint64_t q = S.off[j]; q < S.off[j + 1]; ++q)
    if (S.node[q] == ni->second) {
      I.append_res_wire(*out, S.cpu[q], S.mem[q], I.mem_sw_of(j, q), S.lo[q], S.hi[q], S.g[q], t1, t2, t3, t4, S.w2[q], S.w3[q]);
      return true;
    }
  return false;
}

void GpuNodeSelectionAlgo::ComposeJobToDWire(uint32_t job_id, uint32_t uid, const std::string& partition, const std::string& account,
                                             const std::string& qos, const std::string& name, const std::string& res_wire,
                                             std::string* out, const ArrayTaskIdentity* array_task) {
  Impl::put_u64(*out, 1, job_id);
  Impl::put_u64(*out, 2, uid);
  Impl::put_str(*out, 4, res_wire);   // a message field is written whenever it is set (mutable_res(), CtldPublicDefs.cpp:537-554)
  if (!partition.empty()) Impl::put_str(*out, 5, partition);
  if (!account.empty()) Impl::put_str(*out, 6, account);
  if (!qos.empty()) Impl::put_str(*out, 7, qos);
  if (!name.empty()) Impl::put_str(*out, 9, name);
  if (array_task) {   // optional message field 16: written whenever it is set, even when both scalars are zero
    std::string id;
    Impl::put_u64(id, 1, array_task->array_job_id);
    Impl::put_u64(id, 2, array_task->task_id);
    Impl::put_str(*out, 16, id);
  }
}

bool GpuNodeSelectionAlgo::AppendJobToDWire(const PdJobInScheduler& job, uint32_t uid, const std::string& name,
                                            const CranedId& craned_id, std::string* out, const ArrayTaskIdentity* array_task) {
  std::string res;
  if (!AppendResourceInNodeV3Wire(job, craned_id, &res)) return false;
  ComposeJobToDWire(job.job_id, uid, job.partition_id, job.account, job.qos, name, res, out, array_task);
  return true;
}

void GpuNodeSelectionAlgo::WireOfPackedForTest(int64_t cpu_raw, uint64_t mem, uint64_t mem_sw, uint64_t core_lo, uint64_t core_hi,
                                               uint64_t gres, std::string* wire, ResourceInNodeV3* obj, uint64_t core_w2, uint64_t core_w3) const {
  const Impl& I = *impl_;
  std::string t1, t2, t3, t4;
  I.append_res_wire(*wire, cpu_raw, mem, mem_sw, core_lo, core_hi, gres, t1, t2, t3, t4, core_w2, core_w3);
  *obj = ResourceInNodeV3{};
  I.fill_res(*obj, cpu_raw, mem, core_lo, core_hi, gres, core_w2, core_w3);
  obj->memory_sw_bytes = mem_sw;
}

double GpuNodeSelectionAlgo::EmitWireForBench(size_t* records, size_t* bytes) {
  WireBatch wb;
  const auto t0 = std::chrono::steady_clock::now();
  *records = EmitStartedResourcesWire(&wb);
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  *bytes = wb.bytes.size();
  return ms;
}


// ---------------------------------------------------------------------------------------------------------
// Run-limit admission (JobScheduler.cpp:1557-1573 -> AccountMetaContainer.cpp:180-224).  Host work: names -> dense
// indices and maps -> tables (what the reference does with string-keyed hash maps per job); every comparison and
// usage update runs on the device (cns_apply_run_limits).
// ---------------------------------------------------------------------------------------------------------
namespace {
ResourceView UnlimitedTres() {  // DbClient.cpp:420-428
  ResourceView v;
  v.cpu_count = cpu_t::from_raw(CNS_LIM_UNLIMITED_CPU_RAW);
  v.memory_bytes = v.memory_sw_bytes = CNS_LIM_MAX_JOB_MEMORY;
  return v;
}
const char* kLimitReasonStr[] = {"", "QosEntryNotFound", "QosCpuResourceLimit", "QosJobsResourceLimit", "QosWallTimeLimit",
                                 "QosCpuResourceLimit", "QosMemResourceLimit", "QosGresResourceLimit", "PartitionEntryNotFound",
                                 "UserPartitionJobsLimit", "UserPartitionWallTimeLimit", "AccPartitionJobsLimit",
                                 "AccPartitionWallTimeLimit", "PartitionCpuResourceLimit", "PartitionMemResourceLimit",
                                 "PartitionGresResourceLimit"};
}  // namespace

Qos::Qos() : max_tres(UnlimitedTres()), max_tres_per_user(UnlimitedTres()), max_tres_per_account(UnlimitedTres()) {}
PartitionResourceLimit::PartitionResourceLimit() : max_tres(UnlimitedTres()) {}

void GpuNodeSelectionAlgo::CheckAndMallocMetaResource(AccountMetaSnapshot& meta,
                                                      const std::vector<std::unique_ptr<PdJobInScheduler>>& pending_jobs,
                                                      std::vector<std::string>& results) {
  Impl& I = *impl_;
  results.assign(pending_jobs.size(), std::string());
  auto fail_all = [&](int st, const std::string& msg) {
    status_ = st; error_ = msg;
    for (size_t i = 0; i < pending_jobs.size(); ++i) results[i] = pending_jobs[i]->reason.empty() ? "GpuEngineError" : pending_jobs[i]->reason;
  };
  if (!I.h) return fail_all(status_ ? status_ : CNS_ERR_NO_DEVICE, error_);
  if (!I.have_snapshot) return fail_all(CNS_ERR_STATE, "CheckAndMallocMetaResource before SetClusterSnapshot / NodeSelect");

  // ---- dense indices (sorted names: deterministic) ----
  auto index_of = [](const auto& m) {
    std::map<std::string, uint32_t> ix;
    for (const auto& kv : m) ix.emplace(kv.first, 0);
    uint32_t n = 0;
    for (auto& kv : ix) kv.second = n++;
    return ix;
  };
  const std::map<std::string, uint32_t> qos_ix = index_of(meta.qos), acct_ix = index_of(meta.account_parent), user_ix = index_of(meta.user_accounts);
  std::map<std::pair<std::string, std::string>, uint32_t> ua_ix;  // (user, account) pairs of User::account_to_attrs_map
  for (const auto& [u, accts] : meta.user_accounts)
    for (const auto& [a, lims] : accts) ua_ix.emplace(std::make_pair(u, a), 0);
  { uint32_t n = 0; for (auto& kv : ua_ix) kv.second = n++; }
  const uint32_t Q = (uint32_t)qos_ix.size(), A = (uint32_t)acct_ix.size(), U = (uint32_t)user_ix.size(), UA = (uint32_t)ua_ix.size();
  const uint32_t Pn = (uint32_t)I.part_idx.size();
  std::vector<std::string> part_name(Pn);
  for (const auto& [name, ix] : I.part_idx) part_name[ix] = name;
  if (!Q) return fail_all(CNS_ERR_INVALID_ARG, "AccountMetaSnapshot without QoS");

  auto to_tres = [&](const ResourceView& v) {
    cns_tres t{};
    t.cpu_raw = v.cpu_count.raw;
    t.mem = v.memory_bytes;
    for (const auto& [name, gc] : v.gres_map) {
      auto nit = I.name_id.find(name);
      if (nit == I.name_id.end()) continue;
      t.name_mask |= 1u << nit->second;
      t.name_total[nit->second] = gc.total;
      for (const auto& [type, cnt] : gc.specified) {
        const int c = I.class_of(name, type);
        if (c < 0) continue;
        t.class_mask |= 1u << c;
        t.class_count[c] = cnt;
      }
    }
    return t;
  };
  auto to_usage = [&](const MetaResource& m) {
    cns_usage u{};
    u.cpu_raw = m.resource.cpu_count.raw;
    u.mem = m.resource.memory_bytes;
    u.wall_sec = m.wall_time;
    u.jobs_count = m.jobs_count;
    for (const auto& [name, gc] : m.resource.gres_map) {
      auto nit = I.name_id.find(name);
      if (nit == I.name_id.end()) continue;
      u.name_total[nit->second] = gc.total;
      for (const auto& [type, cnt] : gc.specified) {
        const int c = I.class_of(name, type);
        if (c >= 0) u.class_count[c] = cnt;
      }
    }
    return u;
  };
  auto from_usage = [&](const cns_usage& u) {
    MetaResource m;
    m.resource.cpu_count = cpu_t::from_raw(u.cpu_raw);
    m.resource.memory_bytes = u.mem;
    m.wall_time = u.wall_sec;
    m.jobs_count = u.jobs_count;
    for (const auto& [name, id] : I.name_id)
      if (u.name_total[id]) m.resource.gres_map[name].total = u.name_total[id];
    for (size_t c = 0; c < I.classes.size(); ++c)
      if (u.class_count[c]) m.resource.gres_map[I.classes[c].first].specified[I.classes[c].second] = u.class_count[c];
    return m;
  };
  // the admission changes resource, jobs_count and wall_time of an entry; submit_jobs_count (AccountMetaContainer.h:33) is
  // submit-time bookkeeping and stays what the caller handed in
  auto put_usage = [&](MetaResource& dst, const cns_usage& u) {
    const uint32_t keep = dst.submit_jobs_count;
    dst = from_usage(u);
    dst.submit_jobs_count = keep;
  };

  // ---- limits ----
  std::vector<cns_qos_limits> qos(Q);
  for (const auto& [name, ix] : qos_ix) {
    const Qos& s = meta.qos.at(name);
    cns_qos_limits& d = qos[ix];
    d.max_jobs_per_user = s.max_jobs_per_user; d.max_jobs_per_account = s.max_jobs_per_account; d.max_jobs = s.max_jobs;
    d.max_cpus_per_user_raw = s.max_cpus_per_user.raw; d.max_wall_sec = s.max_wall;
    d.max_tres = to_tres(s.max_tres); d.max_tres_per_user = to_tres(s.max_tres_per_user); d.max_tres_per_account = to_tres(s.max_tres_per_account);
  }
  std::vector<uint32_t> parent(A, CNS_LIM_NONE);
  for (const auto& [name, ix] : acct_ix) {
    const std::string& p = meta.account_parent.at(name);
    if (!p.empty()) { auto it = acct_ix.find(p); if (it != acct_ix.end()) parent[ix] = it->second; }
  }
  std::vector<cns_part_limit> plims;
  auto add_plim = [&](const PartitionResourceLimit& s) {
    cns_part_limit d{};
    d.max_jobs = s.max_jobs; d.max_wall_sec = s.max_wall; d.max_tres = to_tres(s.max_tres);
    plims.push_back(d);
    return (uint32_t)plims.size() - 1;
  };
  std::vector<uint32_t> upl((size_t)UA * Pn, CNS_LIM_NONE), apl((size_t)A * Pn, CNS_LIM_NONE);
  for (const auto& [key, x] : ua_ix)
    for (const auto& [pname, lim] : meta.user_accounts.at(key.first).at(key.second)) {
      auto pit = I.part_idx.find(pname);
      if (pit != I.part_idx.end()) upl[(size_t)x * Pn + pit->second] = add_plim(lim);
    }
  for (const auto& [aname, lims] : meta.account_partition_limits) {
    auto ait = acct_ix.find(aname);
    if (ait == acct_ix.end()) continue;
    for (const auto& [pname, lim] : lims) {
      auto pit = I.part_idx.find(pname);
      if (pit != I.part_idx.end()) apl[(size_t)ait->second * Pn + pit->second] = add_plim(lim);
    }
  }
  // ---- usage maps -> tables (an absent map entry = exists 0) ----
  std::vector<cns_usage> uq((size_t)U * Q), up((size_t)UA * Pn), aq((size_t)A * Q), ap((size_t)A * Pn), qu(Q);
  std::vector<uint8_t> uqe(uq.size(), 0), upe(up.size(), 0), aqe(aq.size(), 0), ape(ap.size(), 0);
  for (const auto& [uname, stat] : meta.user_meta) {
    auto uit = user_ix.find(uname);
    if (uit == user_ix.end()) continue;
    for (const auto& [qname, m] : stat.qos_to_resource_map) {
      auto qit = qos_ix.find(qname);
      if (qit != qos_ix.end()) { uq[(size_t)uit->second * Q + qit->second] = to_usage(m); uqe[(size_t)uit->second * Q + qit->second] = 1; }
    }
    for (const auto& [aname, pm] : stat.account_to_partition_to_resource_map) {
      auto x = ua_ix.find({uname, aname});
      if (x == ua_ix.end()) continue;
      for (const auto& [pname, m] : pm) {
        auto pit = I.part_idx.find(pname);
        if (pit != I.part_idx.end()) { up[(size_t)x->second * Pn + pit->second] = to_usage(m); upe[(size_t)x->second * Pn + pit->second] = 1; }
      }
    }
  }
  for (const auto& [aname, stat] : meta.account_meta) {
    auto ait = acct_ix.find(aname);
    if (ait == acct_ix.end()) continue;
    for (const auto& [qname, m] : stat.qos_to_resource_map) {
      auto qit = qos_ix.find(qname);
      if (qit != qos_ix.end()) { aq[(size_t)ait->second * Q + qit->second] = to_usage(m); aqe[(size_t)ait->second * Q + qit->second] = 1; }
    }
    for (const auto& [pname, m] : stat.partition_to_resource_map) {
      auto pit = I.part_idx.find(pname);
      if (pit != I.part_idx.end()) { ap[(size_t)ait->second * Pn + pit->second] = to_usage(m); ape[(size_t)ait->second * Pn + pit->second] = 1; }
    }
  }
  for (const auto& [qname, m] : meta.qos_meta) {
    auto qit = qos_ix.find(qname);
    if (qit != qos_ix.end()) qu[qit->second] = to_usage(m);
  }
  cns_limit_tables t{};
  t.num_users = U; t.num_user_accts = UA; t.num_accounts = A; t.num_qos = Q; t.num_partitions = Pn; t.num_part_limits = (uint32_t)plims.size();
  t.qos = qos.data(); t.acct_parent = parent.data(); t.part_limits = plims.data();
  t.user_part_limit = upl.data(); t.acct_part_limit = apl.data();
  t.user_qos = uq.data(); t.user_qos_exists = uqe.data(); t.user_part = up.data(); t.user_part_exists = upe.data();
  t.acct_qos = aq.data(); t.acct_qos_exists = aqe.data(); t.acct_part = ap.data(); t.acct_part_exists = ape.data();
  t.qos_usage = qu.data();
  int st = cns_set_run_limits(I.h, &t);
  if (st != 0) return fail_all(st, cns_last_error(I.h));

  // ---- the pending vector: keys, and the lookups the reference fails on before CheckRunLimits_ ----
  const size_t J = pending_jobs.size();
  std::vector<uint64_t> sel(J, 0);
  std::vector<uint32_t> user(J, 0), ua(J, 0), acct(J, 0), qosv(J, 0), part(J, 0);
  std::vector<int64_t> tl(J, 0);
  std::vector<uint8_t> skip(J, 0);
  if (I.last_index.size() != I.last_ord.size()) {
    I.last_index.clear();
    I.last_index.reserve(I.last_ord.size());
    for (size_t j = 0; j < I.last_ord.size(); ++j) I.last_index[I.last_ord[j]] = j;
  }
  for (size_t i = 0; i < J; ++i) {
    const PdJobInScheduler& p = *pending_jobs[i];
    auto li = I.last_index.find(&p);
    if (!p.reason.empty() || li == I.last_index.end()) { skip[i] = 1; results[i] = p.reason; continue; }   // :1507-1510
    sel[i] = li->second;
    tl[i] = p.time_limit;
    const char* err = nullptr;
    auto uit = user_ix.find(p.username);
    auto qit = qos_ix.find(p.qos);
    auto ait = acct_ix.find(p.account);
    auto xit = ua_ix.find({p.username, p.account});
    auto pit = I.part_idx.find(p.partition_id);
    if (uit == user_ix.end()) err = "InvalidUser";                                    // :186-191
    else if (ait == acct_ix.end()) err = "InvalidAccount";                            // :193-199,958-963
    else if (qit == qos_ix.end()) err = "InvalidQOS";                                 // :201-202
    else if (!meta.user_meta.count(p.username)) err = "UserMetaNotFound";            // :895-899
    if (!err && uit != user_ix.end() && ait != acct_ix.end() && qit != qos_ix.end()) {
      // CheckRunLimits_ looks things up in this order (AccountMetaContainer.cpp:895-928): user meta, every account of the
      // chain, qos meta, then the user's account list — the first miss names the reason
      for (std::string a = p.account; !a.empty(); a = meta.account_parent.count(a) ? meta.account_parent.at(a) : std::string())
        if (!meta.account_meta.count(a)) { err = "AccountMetaNotFound"; break; }     // :901-907
      if (!err && !meta.qos_meta.count(p.qos)) err = "QosMetaNotFound";               // :909-913
      if (!err && xit == ua_ix.end()) err = "UserAccountMismatch";                    // :920-928
      if (!err && pit == I.part_idx.end()) err = "Partition Not Found";
    }
    if (err) { skip[i] = 1; results[i] = err; continue; }
    user[i] = uit->second; ua[i] = xit->second; acct[i] = ait->second; qosv[i] = qit->second; part[i] = pit->second;
  }
  cns_limit_job_soa lj{};
  lj.num_jobs = J; lj.select_index = sel.data(); lj.user = user.data(); lj.user_acct = ua.data(); lj.account = acct.data();
  lj.qos = qosv.data(); lj.partition = part.data(); lj.time_limit_sec = tl.data(); lj.skip = skip.data();
  std::vector<uint8_t> reason(J + 1, 0);
  uint64_t admitted = 0;
  st = cns_apply_run_limits(I.h, &lj, reason.data(), &admitted);
  if (st != 0) return fail_all(st, cns_last_error(I.h));
  for (size_t i = 0; i < J; ++i)
    if (!skip[i]) results[i] = reason[i] < 16 ? kLimitReasonStr[reason[i]] : "GpuEngineError";

  // ---- DoMallocResource_'s result back into the caller's maps ----
  st = cns_get_usage(I.h, uq.data(), uqe.data(), up.data(), upe.data(), aq.data(), aqe.data(), ap.data(), ape.data(), qu.data());
  if (st != 0) { status_ = st; error_ = cns_last_error(I.h); return; }
  for (const auto& [uname, uix] : user_ix) {
    for (const auto& [qname, qix] : qos_ix)
      if (uqe[(size_t)uix * Q + qix]) put_usage(meta.user_meta[uname].qos_to_resource_map[qname], uq[(size_t)uix * Q + qix]);
  }
  for (const auto& [key, x] : ua_ix)
    for (uint32_t pp = 0; pp < Pn; ++pp)
      if (upe[(size_t)x * Pn + pp]) put_usage(meta.user_meta[key.first].account_to_partition_to_resource_map[key.second][part_name[pp]], up[(size_t)x * Pn + pp]);
  for (const auto& [aname, aix] : acct_ix) {
    for (const auto& [qname, qix] : qos_ix)
      if (aqe[(size_t)aix * Q + qix]) put_usage(meta.account_meta[aname].qos_to_resource_map[qname], aq[(size_t)aix * Q + qix]);
    for (uint32_t pp = 0; pp < Pn; ++pp)
      if (ape[(size_t)aix * Pn + pp]) put_usage(meta.account_meta[aname].partition_to_resource_map[part_name[pp]], ap[(size_t)aix * Pn + pp]);
  }
  for (const auto& [qname, qix] : qos_ix)
    if (meta.qos_meta.count(qname)) put_usage(meta.qos_meta[qname], qu[qix]);
  status_ = 0;
  error_.clear();
}


// ---------------------------------------------------------------------------------------------------------
// Step scheduling (JobScheduler.cpp:1992-2001 -> CtldPublicDefs.cpp:2038-2159), all jobs in one device call.
// ---------------------------------------------------------------------------------------------------------
void GpuNodeSelectionAlgo::SchedulePendingSteps(std::vector<JobStepQueue>& jobs) {
  Impl& I = *impl_;
  if (!I.h) { status_ = status_ ? status_ : CNS_ERR_NO_DEVICE; return; }
  if (!I.have_snapshot) { status_ = CNS_ERR_STATE; error_ = "SchedulePendingSteps before SetClusterSnapshot"; return; }
  std::vector<uint32_t> noff{0}, nidx, soff{0};
  std::vector<int64_t> acpu;
  std::vector<uint64_t> amem, alo, ahi, ag, aw2, aw3;
  std::vector<StepInScheduler*> flat;
  bool steps_overflow = false;   // (of THIS pass only: it says nothing about the next NodeSelect)
  for (const JobStepQueue& jq : jobs) {
    std::vector<std::pair<uint32_t, const ResourceInNodeV3*>> nodes;   // canonical walk order: ascending dense index
    if (jq.step_res_avail)
      for (const auto& [cid, res] : *jq.step_res_avail) {
        auto it = I.node_idx.find(cid);
        if (it != I.node_idx.end()) nodes.emplace_back(it->second, &res);
      }
    std::sort(nodes.begin(), nodes.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    for (const auto& [n, res] : nodes) {
      nidx.push_back(n);
      acpu.push_back(res->cpu_set.cpu_count.raw);
      amem.push_back(res->memory_bytes);
      uint64_t lo, hi, x2, x3;
      steps_overflow |= I.core_masks(res->cpu_set.core_ids, lo, hi, x2, x3);
      alo.push_back(lo); ahi.push_back(hi); aw2.push_back(x2); aw3.push_back(x3);
      ag.push_back(I.gres_mask(res->gres));
    }
    noff.push_back((uint32_t)nidx.size());
    for (StepInScheduler* s : jq.pending_steps) flat.push_back(s);
    soff.push_back((uint32_t)flat.size());
  }
  if (steps_overflow) {
    status_ = CNS_ERR_UNSUPPORTED;
    error_ = "a job's step_res_avail holds a core id >= 256 (the engine keeps core ids in four 64-bit masks); keep JobInCtld::SchedulePendingSteps for this pass";
    return;
  }
  const size_t S = flat.size();
  std::vector<int64_t> ncpu(S), tcpu(S);
  std::vector<uint64_t> nmem(S), tmem(S);
  std::vector<uint8_t> ngt(S * CNS_MAX_GRES_NAMES, 0), ngs(S * CNS_MAX_GRES_CLASSES, 0), tgt(S * CNS_MAX_GRES_NAMES, 0), tgs(S * CNS_MAX_GRES_CLASSES, 0);
  std::vector<uint32_t> k(S), nt(S), tmin(S), tmax(S), ioff{0}, inn, eoff{0}, enn;
  auto pack_gres = [&](const ResourceView& v, uint8_t* gt, uint8_t* gs) {
    for (const auto& [name, gc] : v.gres_map) {
      auto nit = I.name_id.find(name);
      if (nit == I.name_id.end()) { if (gc.total || !gc.specified.empty()) gt[0] = 255; continue; }   // name absent everywhere: never fits
      gt[nit->second] = (uint8_t)std::min<uint64_t>(gc.total, 255);
      for (const auto& [type, cnt] : gc.specified) {
        const int c = I.class_of(name, type);
        if (c < 0) { if (cnt) gt[nit->second] = 255; continue; }
        gs[c] = (uint8_t)std::min<uint64_t>(cnt, 127);
      }
    }
  };
  for (size_t s = 0; s < S; ++s) {
    const StepInScheduler& st = *flat[s];
    ncpu[s] = st.req_node_res_view.cpu_count.raw; nmem[s] = st.req_node_res_view.memory_bytes;
    tcpu[s] = st.req_task_res_view.cpu_count.raw; tmem[s] = st.req_task_res_view.memory_bytes;
    pack_gres(st.req_node_res_view, &ngt[s * CNS_MAX_GRES_NAMES], &ngs[s * CNS_MAX_GRES_CLASSES]);
    pack_gres(st.req_task_res_view, &tgt[s * CNS_MAX_GRES_NAMES], &tgs[s * CNS_MAX_GRES_CLASSES]);
    k[s] = st.node_num; nt[s] = st.ntasks; tmin[s] = st.ntasks_per_node_min; tmax[s] = st.ntasks_per_node_max;
    for (const auto& n : st.included_nodes) { auto it = I.node_idx.find(n); inn.push_back(it == I.node_idx.end() ? 0xFFFFFFFEu : it->second); }
    ioff.push_back((uint32_t)inn.size());
    for (const auto& n : st.excluded_nodes) { auto it = I.node_idx.find(n); if (it != I.node_idx.end()) enn.push_back(it->second); }
    eoff.push_back((uint32_t)enn.size());
  }
  uint64_t places = 0, tasks = 0;
  for (size_t s = 0; s < S; ++s) { places += k[s]; tasks += nt[s]; }
  const size_t Nn = nidx.size();
  std::vector<uint8_t> sch(S + 1);
  std::vector<uint64_t> poff(S + 1), toff(S + 1), o_mem(places + 1), o_lo(places + 1), o_hi(places + 1), o_g(places + 1),
      t_mem(tasks + 1), t_lo(tasks + 1), t_hi(tasks + 1), t_g(tasks + 1), r_mem(Nn + 1), r_lo(Nn + 1), r_hi(Nn + 1), r_g(Nn + 1),
      o_w2(places + 1), o_w3(places + 1), t_w2(tasks + 1), t_w3(tasks + 1), r_w2(Nn + 1), r_w3(Nn + 1);
  std::vector<uint32_t> o_node(places + 1), o_nt(places + 1), t_node(tasks + 1);
  std::vector<int64_t> o_cpu(places + 1), t_cpu(tasks + 1), r_cpu(Nn + 1);
  cns_step_job_soa cj{};
  cj.num_jobs = (uint32_t)jobs.size(); cj.num_nodes = (uint32_t)Nn; cj.node_offsets = noff.data(); cj.node_idx = nidx.data();
  cj.avail_cpu_raw = acpu.data(); cj.avail_mem = amem.data(); cj.avail_core_lo = alo.data(); cj.avail_core_hi = ahi.data();
  cj.avail_gres = ag.data(); cj.step_offsets = soff.data(); cj.avail_core_w2 = aw2.data(); cj.avail_core_w3 = aw3.data();
  cns_step_soa cs{};
  cs.num_steps = (uint32_t)S; cs.node_cpu_raw = ncpu.data(); cs.node_mem = nmem.data(); cs.node_gres_total = ngt.data(); cs.node_gres_spec = ngs.data();
  cs.task_cpu_raw = tcpu.data(); cs.task_mem = tmem.data(); cs.task_gres_total = tgt.data(); cs.task_gres_spec = tgs.data();
  cs.node_num = k.data(); cs.ntasks = nt.data(); cs.ntasks_per_node_min = tmin.data(); cs.ntasks_per_node_max = tmax.data();
  if (inn.empty()) inn.push_back(0);
  if (enn.empty()) enn.push_back(0);
  cs.incl_offsets = ioff.data(); cs.incl_nodes = inn.data(); cs.excl_offsets = eoff.data(); cs.excl_nodes = enn.data();
  cns_step_result_soa co{};
  co.scheduled = sch.data(); co.place_offsets = poff.data(); co.node_idx = o_node.data(); co.node_ntasks = o_nt.data();
  co.node_cpu_raw = o_cpu.data(); co.node_mem = o_mem.data(); co.node_core_lo = o_lo.data(); co.node_core_hi = o_hi.data(); co.node_gres = o_g.data();
  co.task_offsets = toff.data(); co.task_node = t_node.data(); co.task_cpu_raw = t_cpu.data(); co.task_mem = t_mem.data();
  co.task_core_lo = t_lo.data(); co.task_core_hi = t_hi.data(); co.task_gres = t_g.data();
  co.avail_cpu_raw = r_cpu.data(); co.avail_mem = r_mem.data(); co.avail_core_lo = r_lo.data(); co.avail_core_hi = r_hi.data(); co.avail_gres = r_g.data();
  co.node_core_w2 = o_w2.data(); co.node_core_w3 = o_w3.data(); co.task_core_w2 = t_w2.data(); co.task_core_w3 = t_w3.data();
  co.avail_core_w2 = r_w2.data(); co.avail_core_w3 = r_w3.data();
  const int st = cns_schedule_steps(I.h, &cj, &cs, &co, nullptr);
  if (st != 0) { status_ = st; error_ = cns_last_error(I.h); return; }
  status_ = 0;
  error_.clear();
  // ---- write back (:2109-2135) ----
  for (size_t s = 0; s < S; ++s) {
    StepInScheduler& x = *flat[s];
    x.scheduled = sch[s] != 0;
    x.craned_ids.clear(); x.allocated_res.clear(); x.craned_task_map.clear(); x.task_res_map.clear();
    if (!x.scheduled) continue;
    for (uint64_t p = poff[s]; p < poff[s + 1]; ++p) {
      const CranedId& cid = I.node_name[o_node[p]];
      x.craned_ids.push_back(cid);
      x.allocated_res[cid] = I.to_res(o_cpu[p], o_mem[p], o_lo[p], o_hi[p], o_g[p], o_w2[p], o_w3[p]);
    }
    for (uint64_t t = toff[s]; t < toff[s + 1]; ++t) {
      if (t_node[t] == CNS_NODE_NONE) continue;   // ntasks > what was handed out cannot happen for a scheduled step
      const uint32_t tid = (uint32_t)(t - toff[s]);
      x.craned_task_map[I.node_name[t_node[t]]].insert(tid);
      x.task_res_map[tid] = I.to_res(t_cpu[t], t_mem[t], t_lo[t], t_hi[t], t_g[t], t_w2[t], t_w3[t]);
    }
  }
  for (size_t j = 0; j < jobs.size(); ++j) {
    if (!jobs[j].step_res_avail) continue;
    for (uint32_t n = noff[j]; n < noff[j + 1]; ++n) {
      ResourceInNodeV3& dst = (*jobs[j].step_res_avail)[I.node_name[nidx[n]]];
      const uint64_t sw = dst.memory_sw_bytes;
      dst = I.to_res(r_cpu[n], r_mem[n], r_lo[n], r_hi[n], r_g[n], r_w2[n], r_w3[n]);
      dst.memory_sw_bytes = sw;   // not touched by this path in the canonical model
    }
  }
}


// ---------------------------------------------------------------------------------------------------------
// GpuMultiFactorPriority
// ---------------------------------------------------------------------------------------------------------
GpuMultiFactorPriority::GpuMultiFactorPriority(const PriorityConfig& cfg, int device) : cfg_(cfg) {
  cns_config c{};
  c.abi_version = CNS_ABI_VERSION;
  c.device = device;
  status_ = cns_create(&c, &h_);
  if (status_ != 0) error_ = cns_last_error(nullptr);
}
GpuMultiFactorPriority::~GpuMultiFactorPriority() {
  if (h_) cns_destroy(h_);
}

void GpuMultiFactorPriority::GetOrderedJobPtrVec(const TimeSec& now,
                                                 const std::vector<std::unique_ptr<PdJobInScheduler>>& pending_jobs,
                                                 const std::vector<std::unique_ptr<RnJobInScheduler>>& running_jobs,
                                                 size_t limit, std::vector<PdJobInScheduler*>& job_ptr_vec) {
  const size_t J = pending_jobs.size(), R = running_jobs.size();
  job_ptr_vec.clear();
  job_ptr_vec.reserve(J);
  auto keep_input_order = [&](int st, const std::string& msg) {
    status_ = st; error_ = msg;
    for (const auto& j : pending_jobs) job_ptr_vec.push_back(j.get());
  };
  if (!h_) return keep_input_order(status_ ? status_ : CNS_ERR_NO_DEVICE, error_);
  // dense account ids (the reference keys its service-value map by account name, cpp:7667,7744)
  std::unordered_map<std::string, uint32_t> acc_id;
  auto id_of = [&](const std::string& a) { return acc_id.emplace(a, (uint32_t)acc_id.size()).first->second; };
  std::vector<int64_t> submit(J), cpu(J), r_start(R), r_cpu(R);
  std::vector<uint32_t> qos(J), part(J), nn(J), acc(J), r_qos(R), r_part(R), r_nn(R), r_acc(R);
  std::vector<uint64_t> mem(J), r_mem(R);
  std::vector<double> cached(J);
  for (size_t i = 0; i < J; ++i) {
    const PdJobInScheduler& p = *pending_jobs[i];
    submit[i] = p.submit_time; qos[i] = p.qos_priority; part[i] = p.partition_priority; nn[i] = p.node_num;
    cpu[i] = p.req_total_res_view.cpu_count.raw; mem[i] = p.req_total_res_view.memory_bytes;
    acc[i] = id_of(p.account); cached[i] = p.priority;
  }
  for (size_t i = 0; i < R; ++i) {
    const RnJobInScheduler& r = *running_jobs[i];
    r_start[i] = r.start_time; r_qos[i] = r.qos_priority; r_part[i] = r.partition_priority;
    r_nn[i] = r.node_num ? r.node_num : (uint32_t)r.allocated_res.size();
    r_cpu[i] = r.allocated_res_view.cpu_count.raw; r_mem[i] = r.allocated_res_view.memory_bytes;
    r_acc[i] = id_of(r.account);
  }
  cns_priority_config c{};
  c.max_age_sec = cfg_.MaxAge; c.weight_age = cfg_.WeightAge; c.weight_fair_share = cfg_.WeightFairShare;
  c.weight_job_size = cfg_.WeightJobSize; c.weight_partition = cfg_.WeightPartition; c.weight_qos = cfg_.WeightQoS;
  c.favor_small = cfg_.FavorSmall ? 1u : 0u;
  cns_prio_pending_soa pd{};
  pd.num_jobs = (uint32_t)J; pd.submit_sec = submit.data(); pd.qos_priority = qos.data(); pd.partition_priority = part.data();
  pd.node_num = nn.data(); pd.total_cpu_raw = cpu.data(); pd.total_mem = mem.data(); pd.account = acc.data();
  pd.cached_priority = cached.data();
  cns_prio_running_soa rn{};
  rn.num_jobs = (uint32_t)R; rn.start_sec = r_start.data(); rn.qos_priority = r_qos.data(); rn.partition_priority = r_part.data();
  rn.node_num = r_nn.data(); rn.alloc_cpu_raw = r_cpu.data(); rn.alloc_mem = r_mem.data(); rn.account = r_acc.data();
  std::vector<uint32_t> order(J + 1);
  std::vector<double> prio(J + 1);
  uint64_t nord = 0;
  const int st = cns_priority_order(h_, now, &c, (uint32_t)acc_id.size(), &pd, R ? &rn : nullptr, limit, order.data(),
                                    prio.data(), &nord);
  if (st != 0) return keep_input_order(st, cns_last_error(h_));
  status_ = 0;
  error_.clear();
  for (size_t i = 0; i < J; ++i) pending_jobs[i]->priority = prio[i];       // cpp:7616-7618
  for (size_t i = 0; i < J; ++i) {
    PdJobInScheduler* j = pending_jobs[order[i]].get();
    if (i < nord) job_ptr_vec.push_back(j);
    else j->reason = "Priority";                                             // cpp:7625-7630
  }
}

}  // namespace crane
