// ORACLE / TEST INFRASTRUCTURE ONLY — driver of the REFERENCE's own node-selection code.
//
// This translation unit includes, verbatim, the ranges that oracle/ref_build/extract.py slices out of
// /root/reference at build time (oracle/_ref/gen/*.inc, never committed):
//   crane/PublicHeader.h   resource classes            PublicHeader.cpp   the resource algebra
//   JobScheduler.h         namespace Ctld … SchedulerAlgo (NodeState, NodeSelector, LocalScheduler,
//                          EarliestStartSubsetSelector, PreemptSegTree, the priority sorters)
//   JobScheduler.cpp       LocalScheduler::*, SchedulerAlgo::NodeSelect, MultiFactorPriority::*
//   AccountMetaContainer.h MetaResource, MetaResourceStat, class AccountMetaContainer
//   AccountMetaContainer.cpp  CheckAndMallocMetaResource, CheckRunLimits_, the per-entity checks, CheckTres_ / CheckGres_,
//                          DoMallocResource_ (the run-limit admission of the commit loop)
//   CtldPublicDefs.cpp     JobInCtld::SchedulePendingSteps
// against the stand-ins in oracle/ref_build/shim/ (abseil time + containers, fpm::fixed, the CraneCtld
// singletons NodeSelect reads).  It then exposes THE SAME C entry points as the restated oracle
// (oracle/crane_oracle.cpp: ora_select*, ora_get_costs, ora_get_timeline, ora_feasible, ora_binop,
// ora_priority_order, ora_run_limits, ora_schedule_steps), so tests/ can run every scenario through the reference's code and diff it against
// the restatement.  Output: oracle/_ref/libcrane_ref.so (and libcrane_ref_hash.so, see below).
//
// Two flavours (oracle/Makefile):
//   libcrane_ref.so       -DCRANE_REF_CANONICAL: `unordered_map` in the slices is an ORDERED map and `sort` is
//                         `stable_sort` (two macro renames, no slice line is edited).  The reference iterates hash
//                         maps of GRES types (PublicHeader.cpp:569,584) and sorts with an unstable sort
//                         (JobScheduler.cpp:6401,7623; JobScheduler.h:316): results there depend on libstdc++'s hash
//                         order / introsort.  The canonical flavour pins those to SURVEY.md §7's order (ascending
//                         type, stable) — the order the restated oracle and the engine use.
//   libcrane_ref_hash.so  no renames: libstdc++'s real unordered_map / sort.  Agrees with the canonical flavour
//                         wherever the unspecified orders cannot matter (tests/test_ref_pin.py checks that).
#include "shim/crane_shim.h"

#ifdef CRANE_REF_CANONICAL
namespace std {
template <class K, class V, class H = void, class E = void, class A = void>
class crane_ref_ordered_map : public std::map<K, V> {
 public:
  using std::map<K, V>::map;
  void reserve(size_t) {}
};
}  // namespace std
#define unordered_map crane_ref_ordered_map
#define sort stable_sort
#endif

// crane/PublicHeader.h resource classes + PublicHeader.cpp algebra
#include "ph_types.inc"
#include "ph_impl.inc"

#include "shim/crane_shim_ctld.h"

#define private public
#define protected public
#include "js_types.inc"
#undef private
#undef protected
#include "js_impl.inc"

// run-limit admission (AccountMetaContainer) and the step scheduler (JobInCtld::SchedulePendingSteps)
#include "shim/crane_shim_acct.h"
#define unexpected crane_ref_unexpected   // see shim/expected_shim.h
#define private public
#include "amc_types.inc"
#undef private
#include "amc_impl.inc"
#include "steps_impl.inc"
#include "license_impl.inc"   // LicenseManager::CheckLicenseCountSufficient (LicenseManager.cpp)
#undef unexpected

#ifdef CRANE_REF_CANONICAL
#undef unordered_map
#undef sort
#endif

#include <chrono>
#include <cstring>

#include "../../include/crane_gpu/node_select.h"
#include "../../include/crane_gpu/preempt.h"
#include "../../include/crane_gpu/run_limits.h"
#include "../../include/crane_gpu/steps.h"

namespace {

using Ctld::PdJobInScheduler;
using Ctld::RnJobInScheduler;
using Ctld::SchedulerAlgo;
using NodeState = SchedulerAlgo::NodeState;
using LocalScheduler = SchedulerAlgo::LocalScheduler;

struct MaskRes { int64_t cpu = 0; uint64_t mem = 0, clo = 0, chi = 0, gres = 0, c2 = 0, c3 = 0; };

struct Layout {
  cns_gres_layout g{};
  int class_of_bit(int b) const {
    for (uint32_t c = 0; c < g.num_classes; ++c)
      if (b >= g.class_shift[c] && b < g.class_shift[c] + g.class_width[c]) return (int)c;
    return -1;
  }
};

std::string node_name(uint32_t n) { return "n" + std::to_string(n); }
std::string part_name(uint32_t p) { char b[16]; snprintf(b, sizeof b, "p%05u", p); return b; }
std::string resv_name(uint32_t v) { char b[16]; snprintf(b, sizeof b, "r%05u", v); return b; }
std::string qos_name(uint32_t q) { char b[16]; snprintf(b, sizeof b, "q%05u", q); return b; }
std::string gres_name(uint32_t a) { return std::string("a") + char('0' + a); }      // a0..a3, ascending = name id
std::string gres_type(uint32_t c) { return std::string("t") + char('0' + c); }      // t0..t7, ascending = class id
std::string slot_name(int bit) { char b[16]; snprintf(b, sizeof b, "/dev/s%02d", bit); return b; }  // lexicographic = bit order
uint32_t idx_of(const std::string& s) { return (uint32_t)strtoul(s.c_str() + 1, nullptr, 10); }

ResourceInNodeV3 to_ref(const Layout& L, const MaskRes& m) {
  ResourceInNodeV3 r;
  r.GetCpuSet().cpu_count = cpu_t::from_raw_value(m.cpu);
  for (int b = 0; b < 64; ++b) {
    if ((m.clo >> b) & 1) r.GetCpuSet().core_ids.insert((uint32_t)b);
    if ((m.chi >> b) & 1) r.GetCpuSet().core_ids.insert((uint32_t)(64 + b));
    if ((m.c2 >> b) & 1) r.GetCpuSet().core_ids.insert((uint32_t)(128 + b));
    if ((m.c3 >> b) & 1) r.GetCpuSet().core_ids.insert((uint32_t)(192 + b));
  }
  r.SetMemoryBytes(m.mem);
  for (int b = 0; b < 64; ++b)
    if ((m.gres >> b) & 1) {
      const int c = L.class_of_bit(b);
      if (c < 0) throw std::invalid_argument("gres bit outside every class");
      r.GetGres()[gres_name(L.g.class_name[c])][gres_type((uint32_t)c)].insert(slot_name(b));
    }
  return r;
}
MaskRes from_ref(const ResourceInNodeV3& r) {
  MaskRes m;
  m.cpu = r.GetCpuSet().cpu_count.raw_value();
  m.mem = r.GetMemoryBytes();
  for (uint32_t c : r.GetCpuSet().core_ids) {
    if (c < 64) m.clo |= 1ull << c;
    else if (c < 128) m.chi |= 1ull << (c - 64);
    else if (c < 192) m.c2 |= 1ull << (c - 128);
    else if (c < 256) m.c3 |= 1ull << (c - 192);
    else throw std::invalid_argument("core id >= 256");
  }
  for (const auto& [name, tsm] : r.GetGres().name_type_slots_map)
    for (const auto& [type, slots] : tsm.type_slots_map)
      for (const auto& s : slots) m.gres |= 1ull << strtoul(s.c_str() + 6, nullptr, 10);
  return m;
}
ResourceView view_of(const Layout& L, int64_t cpu, uint64_t mem, const uint8_t* gtot, const uint8_t* gspec) {
  ResourceView v;
  v.SetCpuCount(cpu_t::from_raw_value(cpu));
  v.SetMemoryBytes(mem);
  for (uint32_t a = 0; a < CNS_MAX_GRES_NAMES; ++a) {
    std::unordered_map<std::string, uint64_t> spec;   // (in the canonical flavour this token is back to the real one)
    if (gspec)
      for (uint32_t c = 0; c < L.g.num_classes; ++c)
        if (L.g.class_name[c] == a && gspec[c]) spec[gres_type(c)] = gspec[c];
    const uint64_t total = gtot ? gtot[a] : 0;
    if (total == 0 && spec.empty()) continue;
    GresCount gc(total);
    for (const auto& [t, n] : spec) gc.specified[t] = n;
    v.GetGresMap()[gres_name(a)] = std::move(gc);
  }
  return v;
}

// jobs of one kind live in one array so that pointer order == queue order (absl_shim.h, flat_hash_set)
template <class T>
struct JobArena {
  T* base = nullptr;
  size_t n = 0, built = 0;
  std::vector<std::unique_ptr<T>> owners;
  explicit JobArena(size_t count) : n(count) {
    base = static_cast<T*>(::operator new(std::max<size_t>(n, 1) * sizeof(T), std::align_val_t(alignof(T))));
    owners.reserve(n);
  }
  template <class... A> T* make(A&&... a) {
    T* p = new (base + built) T(std::forward<A>(a)...);
    ++built;
    owners.emplace_back(p);
    return p;
  }
  ~JobArena() {
    for (auto& o : owners) o.release();
    for (size_t i = 0; i < built; ++i) base[i].~T();
    ::operator delete(base, std::align_val_t(alignof(T)));
  }
  size_t index_of(const T* p) const { return size_t(p - base); }
};

struct RefRun {
  Layout layout;
  std::vector<std::vector<uint32_t>> part_nodes;
  std::map<std::pair<uint32_t, uint32_t>, double> cost;                   // (partition, node) -> final cost
  std::map<uint32_t, std::vector<std::pair<int64_t, MaskRes>>> timeline;  // real nodes' final time maps
  double seconds = 0;
  uint64_t jobs_ordered = 0;
  std::string error;
};

thread_local std::string g_last_error;

int reason_code(const std::string& r, bool failed) {
  if (r.empty()) return CNS_REASON_NONE;
  if (r == "Priority") return CNS_REASON_PRIORITY;
  if (r == "Resource") return CNS_REASON_RESOURCE;
  if (r == "Resource Reserved") return CNS_REASON_RESOURCE_RESERVED;
  if (r == "Partition Not Found") return CNS_REASON_PARTITION_NOT_FOUND;
  if (r == "Reservation Not Found") return CNS_REASON_RESERVATION_NOT_FOUND;
  if (r == "Preempted") return CNS_REASON_PREEMPTED;
  if (r == "__skipped__") return CNS_REASON_SKIPPED;
  (void)failed;
  throw std::runtime_error("unknown reason string: " + r);
}

}  // namespace

extern "C" {

struct ora_res { int64_t cpu; uint64_t mem, clo, chi, gres, c2, c3; };
struct ora_req { int64_t cpu; uint64_t mem; uint8_t gtot[CNS_MAX_GRES_NAMES]; uint8_t gspec[CNS_MAX_GRES_CLASSES]; };

const char* ref_last_error(void) { return g_last_error.c_str(); }
// 1: the slices were compiled with the two canonicalising renames, 0: libstdc++'s hash order / introsort
int ref_is_canonical(void) {
#ifdef CRANE_REF_CANONICAL
  return 1;
#else
  return 0;
#endif
}

// ---- the reference's resource algebra (algebra argument ignored: there is only the reference's) ----
int ora_feasible(const cns_gres_layout* gl, int, const ora_req* req, const ora_res* avail, ora_res* out) {
  try {
    Layout L; L.g = *gl;
    const ResourceView v = view_of(L, req->cpu, req->mem, req->gtot, req->gspec);
    const ResourceInNodeV3 a = to_ref(L, MaskRes{avail->cpu, avail->mem, avail->clo, avail->chi, avail->gres, avail->c2, avail->c3});
    ResourceInNodeV3 f;
    const bool ok = v.GetFeasibleResourceInNode(a, &f);   // PublicHeader.cpp:519-599
    if (ok) { const MaskRes m = from_ref(f); *out = ora_res{m.cpu, m.mem, m.clo, m.chi, m.gres, m.c2, m.c3}; }
    return ok;
  } catch (const std::exception& e) { g_last_error = e.what(); return -1; }
}
int ora_binop(const cns_gres_layout* gl, int, int op, const ora_res* a, const ora_res* b, ora_res* out) {
  try {
    Layout L; L.g = *gl;
    ResourceInNodeV3 x = to_ref(L, MaskRes{a->cpu, a->mem, a->clo, a->chi, a->gres, a->c2, a->c3});
    const ResourceInNodeV3 y = to_ref(L, MaskRes{b->cpu, b->mem, b->clo, b->chi, b->gres, b->c2, b->c3});
    int ret = 0;
    switch (op) {
#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Wdeprecated-declarations"
      case 0: x.Ckmin(y); break;          // PublicHeader.cpp:815-827
#pragma GCC diagnostic pop
      case 1: x += y; break;              // :781-787
      case 2: x -= y; break;              // :789-796
      case 3: ret = (x <= y); break;      // :886-890
    }
    if (out) { const MaskRes m = from_ref(x); *out = ora_res{m.cpu, m.mem, m.clo, m.chi, m.gres, m.c2, m.c3}; }
    return ret;
  } catch (const std::exception& e) { g_last_error = e.what(); return -1; }
}

// ---- one scheduling cycle through SchedulerAlgo::NodeSelect ------------------------------------------------
// rc: 0 ok; -3 a CRANE_ASSERT / ABSL_ASSERT of the reference failed (message: ref_last_error); -5 the call asks
// for a constant the reference fixes at compile time (kAlgoMaxJobNumPerNode, kAlgoMaxTimeWindow); -1 other.
static int ref_select_impl(const cns_config* cfg, const cns_node_soa* nodes, const cns_running_soa* running,
                           const cns_resv_soa* resv, int64_t now_sec, const cns_job_soa* jobs, cns_placement_soa* out,
                           void** run_out, const cns_preempt_soa* pre, cns_preempt_out* pout) {
  using namespace Ctld;
  if (cfg && ((cfg->max_job_num_per_node && cfg->max_job_num_per_node != 1000) ||
              (cfg->max_time_window_sec && cfg->max_time_window_sec != 7 * 24 * 3600))) {
    g_last_error = "kAlgoMaxJobNumPerNode / kAlgoMaxTimeWindow are compile-time constants of the reference (JobScheduler.h:269-270)";
    return -5;
  }
  auto run = std::make_unique<RefRun>();
  run->layout.g = nodes->gres;
  const Layout& L = run->layout;
  const absl::Time now = absl::FromUnixSeconds(now_sec);

  CranedMetaContainer meta;
  AccountManager accounts;
  LicenseManager licenses;
  RefJobSchedulerStub sched_stub;
  try {
    // ---- g_meta_container: craned metas, partitions, reservations --------------------------------------
    crane_ref::g_arena_slots = std::max<size_t>(nodes->num_nodes, 1);
    meta.craneds.reserve(nodes->num_nodes);
    for (uint32_t n = 0; n < nodes->num_nodes; ++n) {
      auto& cm = meta.craneds.add(node_name(n)).value;
      const bool s = nodes->schedulable ? nodes->schedulable[n] != 0 : true;
      cm.alive = s;   // alive && !drain (JobScheduler.cpp:6595)
      cm.drain = false;
      MaskRes t;
      t.cpu = nodes->cpu_total_raw[n]; t.mem = nodes->mem_total[n];
      t.clo = nodes->core_lo ? nodes->core_lo[n] : 0; t.chi = nodes->core_hi ? nodes->core_hi[n] : 0;
      t.c2 = nodes->core_w2 ? nodes->core_w2[n] : 0; t.c3 = nodes->core_w3 ? nodes->core_w3[n] : 0;
      t.gres = nodes->gres_slots ? nodes->gres_slots[n] : 0;
      cm.res_total = to_ref(L, t);
    }
    run->part_nodes.resize(nodes->num_partitions);
    for (uint32_t p = 0; p < nodes->num_partitions; ++p) {
      auto& pm = meta.partitions.add(part_name(p)).value;
      for (uint32_t i = nodes->part_offsets[p]; i < nodes->part_offsets[p + 1]; ++i) {
        pm.craned_ids.push_back(node_name(nodes->part_nodes[i]));
        run->part_nodes[p].push_back(nodes->part_nodes[i]);
      }
    }
    auto alloc_of = [&](const int64_t* cpu, const uint64_t* mem, const uint64_t* lo, const uint64_t* hi, const uint64_t* g, uint32_t a,
                        const uint64_t* w2, const uint64_t* w3) {
      MaskRes m;
      m.cpu = cpu[a]; m.mem = mem[a]; m.clo = lo ? lo[a] : 0; m.chi = hi ? hi[a] : 0; m.gres = g ? g[a] : 0;
      m.c2 = w2 ? w2[a] : 0; m.c3 = w3 ? w3[a] : 0;
      return to_ref(L, m);
    };
    if (resv)
      for (uint32_t v = 0; v < resv->num_resv; ++v) {
        auto& rm = meta.reservations.add(resv_name(v)).value;
        rm.start_time = absl::FromUnixSeconds(resv->start_sec[v]);
        rm.end_time = absl::FromUnixSeconds(resv->end_sec[v]);
        for (uint32_t a = resv->alloc_offsets[v]; a < resv->alloc_offsets[v + 1]; ++a) {
          rm.craned_ids.push_back(node_name(resv->alloc_node[a]));
          rm.res_total.AddResourceInNode(node_name(resv->alloc_node[a]),
                                         alloc_of(resv->alloc_cpu_raw, resv->alloc_mem, resv->alloc_core_lo, resv->alloc_core_hi, resv->alloc_gres, a, resv->alloc_core_w2, resv->alloc_core_w3));
        }
      }
    // ---- running and pending jobs through the reference's own constructors (JobScheduler.h:75-90,143-170) ----
    const uint32_t R = running ? running->num_jobs : 0;
    const uint64_t J = jobs->num_jobs;
    JobArena<RnJobInScheduler> rn_arena(R);
    JobArena<PdJobInScheduler> pd_arena(J);
    for (uint32_t r = 0; r < R; ++r) {
      JobInCtld j;
      j.job_id = pre && pre->rn_job_id ? pre->rn_job_id[r] : r;
      j.qos = qos_name(pre && pre->rn_qos ? pre->rn_qos[r] : 0);
      j.qos_priority = pre && pre->rn_qos_priority ? pre->rn_qos_priority[r] : 0;
      j.start_time = absl::FromUnixSeconds(pre && pre->rn_start_sec ? pre->rn_start_sec[r] : 0);
      j.end_time = absl::FromUnixSeconds(running->end_sec[r]);
      if (running->reservation && running->reservation[r] != CNS_RESV_NONE) j.reservation = resv_name(running->reservation[r]);
      for (uint32_t a = running->alloc_offsets[r]; a < running->alloc_offsets[r + 1]; ++a)
        j.allocated_res.AddResourceInNode(node_name(running->alloc_node[a]),
                                          alloc_of(running->alloc_cpu_raw, running->alloc_mem, running->alloc_core_lo, running->alloc_core_hi, running->alloc_gres, a, running->alloc_core_w2, running->alloc_core_w3));
      rn_arena.make(&j);
    }
    for (uint64_t i = 0; i < J; ++i) {
      JobInCtld j;
      j.job_id = pre && pre->pd_job_id ? pre->pd_job_id[i] : (uint32_t)i;
      j.partition_id = part_name(jobs->partition[i]);
      if (jobs->reservation && jobs->reservation[i] != CNS_RESV_NONE) j.reservation = resv_name(jobs->reservation[i]);
      j.time_limit = absl::Seconds(jobs->time_limit_sec[i]);
      j.req_node_res_view = view_of(L, jobs->node_cpu_raw ? jobs->node_cpu_raw[i] : 0, jobs->node_mem[i],
                                    jobs->gres_total ? jobs->gres_total + i * CNS_MAX_GRES_NAMES : nullptr,
                                    jobs->gres_spec ? jobs->gres_spec + i * CNS_MAX_GRES_CLASSES : nullptr);
      j.req_task_res_view = view_of(L, jobs->task_cpu_raw[i], jobs->task_mem[i], nullptr, nullptr);
      j.node_num = jobs->node_num[i];
      j.ntasks = jobs->ntasks[i];
      j.ntasks_per_node_min = jobs->ntasks_per_node_min[i];
      j.ntasks_per_node_max = jobs->ntasks_per_node_max[i];
      j.exclusive = jobs->exclusive ? jobs->exclusive[i] != 0 : false;
      if (jobs->incl_offsets)
        for (uint64_t k = jobs->incl_offsets[i]; k < jobs->incl_offsets[i + 1]; ++k) j.included_nodes.insert(node_name(jobs->incl_nodes[k]));
      if (jobs->excl_offsets)
        for (uint64_t k = jobs->excl_offsets[i]; k < jobs->excl_offsets[i + 1]; ++k) j.excluded_nodes.insert(node_name(jobs->excl_nodes[k]));
      j.qos = qos_name(pre && pre->pd_qos ? pre->pd_qos[i] : 0);
      j.qos_priority = pre && pre->pd_qos_priority ? pre->pd_qos_priority[i] : 0;
      j.mandated_priority = pre && pre->pd_priority ? pre->pd_priority[i] : 0.0;
      PdJobInScheduler* pd = pd_arena.make(&j);
      if (jobs->skip && jobs->skip[i]) pd->reason = "__skipped__";   // a reason set before NodeSelect (JobScheduler.cpp:6744)
    }
    // ---- config + singletons --------------------------------------------------------------------------
    g_config = Config{};
    g_config.ScheduledBatchSize = (cfg && cfg->scheduled_batch_size) ? (uint32_t)std::min<uint64_t>(cfg->scheduled_batch_size, UINT32_MAX) : UINT32_MAX;
    g_config.Preempt.PreemptType = (pre && pre->enabled) ? crane::grpc::PreemptType::PREEMPT_QOS : crane::grpc::PreemptType::PREEMPT_NONE;
    if (pre)
      for (uint32_t q = 0; q < pre->num_qos; ++q) {
        auto qos = std::make_unique<Qos>();
        for (uint32_t k = pre->qos_preempt_offsets[q]; k < pre->qos_preempt_offsets[q + 1]; ++k) qos->preempt.push_back(qos_name(pre->qos_preempt[k]));
        accounts.qos_map[qos_name(q)] = std::move(qos);
      }
    g_meta_container = &meta; g_account_manager = &accounts; g_license_manager = &licenses; g_job_scheduler = &sched_stub;

    BasicPriority sorter;
    SchedulerAlgo algo(&sorter);
    if (pre)
      for (uint32_t k = 0; k < pre->num_preempting; ++k) algo.m_preempting_set_.insert(pre->preempting_job_ids[k]);

    // ---- observers of NodeSelect's locals (absl_shim.h, DtorHook) ------------------------------------------
    using absl::ref_detail::DtorHook;
    using ResvInner = absl::flat_hash_map<CranedId, NodeState>;
    std::set<const void*> resv_maps;
    DtorHook<ResvId, std::pair<absl::Time, ResvInner>>::fn = [&](const void*, const ResvId&, std::pair<absl::Time, ResvInner>& v) {
      resv_maps.insert(static_cast<const absl::ref_detail::ArenaMap<CranedId, NodeState>*>(&v.second));
    };
    DtorHook<CranedId, NodeState>::fn = [&](const void* map, const CranedId& id, NodeState& ns) {
      if (resv_maps.count(map)) return;   // a reservation's virtual nodes
      auto& tl = run->timeline[idx_of(id)];
      for (const auto& [t, res] : ns.time_avail_res_map)
        tl.emplace_back(t == absl::InfiniteFuture() ? INT64_MAX : absl::ToUnixSeconds(t), from_ref(res));
    };
    bool in_part_map = false;
    DtorHook<PartitionId, LocalScheduler>::fn = [&](const void*, const PartitionId& pid, LocalScheduler& ls) {
      if (pid.empty() || pid[0] != 'p' || !ls.m_node_selector_) return;   // reservations' schedulers are keyed "r…"
      auto* sel = static_cast<SchedulerAlgo::NodeSelector*>(ls.m_node_selector_.get());
      for (auto& [cid, rater] : sel->m_node_info_map_) run->cost[{idx_of(pid), idx_of(cid)}] = rater.cost;
      (void)in_part_map;
    };
    struct HookReset {
      ~HookReset() {
        DtorHook<ResvId, std::pair<absl::Time, ResvInner>>::fn = nullptr;
        DtorHook<CranedId, NodeState>::fn = nullptr;
        DtorHook<PartitionId, LocalScheduler>::fn = nullptr;
      }
    } hook_reset;

    const auto t0 = std::chrono::steady_clock::now();   // the reference's own bracket, JobScheduler.cpp:1439-1447
    algo.NodeSelect(now, rn_arena.owners, pd_arena.owners);
    run->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    run->jobs_ordered = std::min<uint64_t>(J, g_config.ScheduledBatchSize);

    // ---- results -----------------------------------------------------------------------------------------
    if (out) {
      uint64_t off = 0;
      for (uint64_t i = 0; i < J; ++i) {
        const PdJobInScheduler& job = *pd_arena.owners[i];
        const bool failed = job.reason == "Resource" && job.start_time == absl::Time();   // `ok == false`: start_time left unset (:6767-6768)
        const int rc = reason_code(job.reason, failed);
        const bool placed = !failed && rc != CNS_REASON_SKIPPED && rc != CNS_REASON_PARTITION_NOT_FOUND &&
                            rc != CNS_REASON_RESERVATION_NOT_FOUND && !(rc == CNS_REASON_PRIORITY && job.start_time == absl::Time());
        out->place_offsets[i] = off;
        out->reason[i] = (uint8_t)rc;
        out->start_sec[i] = placed ? absl::ToUnixSeconds(job.start_time) : 0;
        uint64_t k = 0;
        if (placed) {
          std::map<uint32_t, const ResourceInNodeV3*> by_node;   // report sorted by node index
          for (const auto& [cid, res] : job.allocated_res.EachNodeResMap()) by_node[idx_of(cid)] = &res;
          for (const auto& [nid, res] : by_node) {
            const uint64_t q = off + k++;
            const MaskRes m = from_ref(*res);
            out->node_idx[q] = nid;
            out->ntasks[q] = job.craned_id_to_task_num.at(node_name(nid));
            out->cpu_raw[q] = m.cpu; out->mem[q] = m.mem; out->core_lo[q] = m.clo; out->core_hi[q] = m.chi; out->gres[q] = m.gres;
            if (out->core_w2) out->core_w2[q] = m.c2;
            if (out->core_w3) out->core_w3[q] = m.c3;
          }
        }
        for (; k < job.node_num; ++k) {
          const uint64_t q = off + k;
          out->node_idx[q] = CNS_NODE_NONE; out->ntasks[q] = 0; out->cpu_raw[q] = 0; out->mem[q] = 0;
          out->core_lo[q] = out->core_hi[q] = out->gres[q] = 0;
          if (out->core_w2) out->core_w2[q] = 0;
          if (out->core_w3) out->core_w3[q] = 0;
        }
        off += job.node_num;
      }
      out->place_offsets[J] = off;
    }
    if (pre && pout) {
      uint64_t off = 0;
      for (uint64_t i = 0; i < J; ++i) {
        pout->offsets[i] = off;
        for (const auto& v : pd_arena.owners[i]->preempted_jobs) {
          if (off >= pout->capacity) return -7;
          if (std::holds_alternative<PdJobInScheduler*>(v)) pout->preempted[off++] = (uint32_t)pd_arena.index_of(std::get<PdJobInScheduler*>(v)) | CNS_PREEMPT_REF_PENDING;
          else pout->preempted[off++] = (uint32_t)rn_arena.index_of(std::get<RnJobInScheduler*>(v));
        }
      }
      pout->offsets[J] = off;
      pout->num_cancelled = 0;
      for (job_id_t id : sched_stub.cancelled) { if (pout->num_cancelled >= pout->cancel_capacity) return -7; pout->cancelled_job_ids[pout->num_cancelled++] = id; }
      pout->num_preempting = 0;
      for (job_id_t id : algo.m_preempting_set_) { if (pout->num_preempting >= pout->preempting_capacity) return -7; pout->preempting_job_ids[pout->num_preempting++] = id; }
    }
  } catch (const crane_ref::RefAssertion& e) {
    g_last_error = std::string("reference assertion failed: ") + e.what();
    return -3;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return -1;
  }
  if (run_out) *run_out = run.release();
  return 0;
}

int ora_select(const cns_config* cfg, const cns_node_soa* nodes, const cns_running_soa* running, int64_t now,
               const cns_job_soa* jobs, cns_placement_soa* out, int, void** run_out) {
  return ref_select_impl(cfg, nodes, running, nullptr, now, jobs, out, run_out, nullptr, nullptr);
}
int ora_select_resv(const cns_config* cfg, const cns_node_soa* nodes, const cns_running_soa* running, const cns_resv_soa* resv,
                    int64_t now, const cns_job_soa* jobs, cns_placement_soa* out, int, void** run_out) {
  return ref_select_impl(cfg, nodes, running, resv, now, jobs, out, run_out, nullptr, nullptr);
}
int ora_select_preempt(const cns_config* cfg, const cns_node_soa* nodes, const cns_running_soa* running, const cns_resv_soa* resv,
                       int64_t now, const cns_job_soa* jobs, const cns_preempt_soa* pre, cns_placement_soa* out,
                       cns_preempt_out* pout, int, void** run_out) {
  return ref_select_impl(cfg, nodes, running, resv, now, jobs, out, run_out, pre, pout);
}

double ora_seconds(void* h) { return static_cast<RefRun*>(h)->seconds; }
uint64_t ora_jobs_ordered(void* h) { return static_cast<RefRun*>(h)->jobs_ordered; }
int ora_get_costs(void* h, double* cost_by_part_slot) {
  auto* run = static_cast<RefRun*>(h);
  size_t q = 0;
  for (uint32_t p = 0; p < run->part_nodes.size(); ++p)
    for (uint32_t n : run->part_nodes[p]) {
      auto it = run->cost.find({p, n});
      cost_by_part_slot[q++] = it == run->cost.end() ? 0.0 : it->second;
    }
  return 0;
}
int ora_get_timeline(void* h, uint32_t node, uint32_t capacity, uint32_t* len, int64_t* t, int64_t* cpu_raw, uint64_t* mem,
                     uint64_t* core_lo, uint64_t* core_hi, uint64_t* gres) {
  auto* run = static_cast<RefRun*>(h);
  auto it = run->timeline.find(node);
  if (it == run->timeline.end()) { *len = 0; return 0; }
  *len = (uint32_t)it->second.size();
  uint32_t i = 0;
  for (const auto& [time, m] : it->second) {
    if (i >= capacity) break;
    t[i] = time; cpu_raw[i] = m.cpu; mem[i] = m.mem; core_lo[i] = m.clo; core_hi[i] = m.chi; gres[i] = m.gres;
    ++i;
  }
  return 0;
}
int ora_get_timeline_cores(void* h, uint32_t node, uint32_t capacity, uint64_t* core_w2, uint64_t* core_w3) {
  auto* run = static_cast<RefRun*>(h);
  auto it = run->timeline.find(node);
  if (it == run->timeline.end()) return 0;
  uint32_t i = 0;
  for (const auto& [time, m] : it->second) {
    if (i >= capacity) break;
    core_w2[i] = m.c2; core_w3[i] = m.c3;
    ++i;
  }
  return 0;
}
void ora_free(void* h) { delete static_cast<RefRun*>(h); }

// ---- MultiFactorPriority::GetOrderedJobPtrVec (JobScheduler.cpp:7606-7819), arrays as oracle/crane_oracle.cpp ----
int ora_priority_order(int64_t now_sec, uint64_t max_age, uint32_t w_age, uint32_t w_fair, uint32_t w_size, uint32_t w_part,
                       uint32_t w_qos, uint32_t favor_small, uint32_t num_accounts, uint32_t J, const int64_t* submit,
                       const uint32_t* qos, const uint32_t* part, const uint32_t* node_num, const int64_t* cpu_raw,
                       const uint64_t* mem, const uint32_t* account, const double* cached, uint32_t R,
                       const int64_t* r_start, const uint32_t* r_qos, const uint32_t* r_part, const uint32_t* r_node_num,
                       const int64_t* r_cpu_raw, const uint64_t* r_mem, const uint32_t* r_account, uint32_t* order_out,
                       double* prio_out) {
  using namespace Ctld;
  try {
    for (uint32_t i = 0; i < J; ++i) if (account[i] >= num_accounts) return -1;
    for (uint32_t i = 0; i < R; ++i) if (r_account[i] >= num_accounts) return -1;
    g_config = Config{};
    g_config.PriorityConfig.MaxAge = max_age;
    g_config.PriorityConfig.WeightAge = w_age; g_config.PriorityConfig.WeightFairShare = w_fair;
    g_config.PriorityConfig.WeightJobSize = w_size; g_config.PriorityConfig.WeightPartition = w_part;
    g_config.PriorityConfig.WeightQoS = w_qos; g_config.PriorityConfig.FavorSmall = favor_small != 0;
    auto acct = [](uint32_t a) { char b[16]; snprintf(b, sizeof b, "acc%06u", a); return std::string(b); };
    Layout L;
    JobArena<RnJobInScheduler> rn_arena(R);
    JobArena<PdJobInScheduler> pd_arena(J);
    for (uint32_t r = 0; r < R; ++r) {
      JobInCtld j;
      j.job_id = r; j.start_time = absl::FromUnixSeconds(r_start[r]); j.qos_priority = r_qos[r]; j.partition_priority = r_part[r];
      j.account = acct(r_account[r]);
      j.allocated_res_view = view_of(L, r_cpu_raw[r], r_mem[r], nullptr, nullptr);
      RnJobInScheduler* rn = rn_arena.make(&j);
      rn->node_num = r_node_num[r];   // not set by the constructor (JobScheduler.h:75-90); the reference reads it at :7690
    }
    for (uint32_t i = 0; i < J; ++i) {
      JobInCtld j;
      j.job_id = i; j.submit_time = absl::FromUnixSeconds(submit[i]); j.qos_priority = qos[i]; j.partition_priority = part[i];
      j.node_num = node_num[i]; j.account = acct(account[i]);
      j.req_total_res_view = view_of(L, cpu_raw[i], mem[i], nullptr, nullptr);
      j.mandated_priority = cached ? cached[i] : 0.0;
      pd_arena.make(&j);
    }
    MultiFactorPriority sorter;
    std::vector<PdJobInScheduler*> vec;
    sorter.GetOrderedJobPtrVec(absl::FromUnixSeconds(now_sec), pd_arena.owners, rn_arena.owners, (size_t)J, vec);
    for (uint32_t i = 0; i < J; ++i) { order_out[i] = (uint32_t)pd_arena.index_of(vec[i]); }
    for (uint32_t i = 0; i < J; ++i) prio_out[i] = pd_arena.owners[i]->priority;
    return 0;
  } catch (const std::exception& e) { g_last_error = e.what(); return -2; }
}

// ---- AccountMetaContainer::CheckAndMallocMetaResource over the commit loop (JobScheduler.cpp:1492-1573) -----------
// Same arguments as oracle/crane_oracle.cpp's ora_run_limits.  reason_out[i] = index of the reference's reason STRING
// in the table of include/crane_gpu/run_limits.h (first match: the reference returns "QosCpuResourceLimit" both for
// max_cpus_per_user, code 2, and for a max_tres* cpu count, code 5 — compare strings, tests/test_ref_pin.py does).
namespace {
const char* const kLimitReasonStr[16] = {"", "QosEntryNotFound", "QosCpuResourceLimit", "QosJobsResourceLimit", "QosWallTimeLimit",
                                         "QosCpuResourceLimit", "QosMemResourceLimit", "QosGresResourceLimit", "PartitionEntryNotFound",
                                         "UserPartitionJobsLimit", "UserPartitionWallTimeLimit", "AccPartitionJobsLimit",
                                         "AccPartitionWallTimeLimit", "PartitionCpuResourceLimit", "PartitionMemResourceLimit",
                                         "PartitionGresResourceLimit"};
std::string user_name(uint32_t u) { char b[16]; snprintf(b, sizeof b, "u%06u", u); return b; }
std::string acct_name(uint32_t a) { char b[16]; snprintf(b, sizeof b, "c%06u", a); return b; }

ResourceView limit_view(const Layout& L, const cns_tres& t) {   // a LIMIT: entries exist where the masks say so
  ResourceView v;
  v.SetCpuCount(cpu_t::from_raw_value(t.cpu_raw));
  v.SetMemoryBytes(t.mem);
  for (uint32_t n = 0; n < CNS_MAX_GRES_NAMES; ++n)
    if (t.name_mask >> n & 1) v.GetGresMap()[gres_name(n)].total = t.name_total[n];
  for (uint32_t g = 0; g < L.g.num_classes; ++g)
    if ((t.class_mask >> g & 1) && (t.name_mask >> L.g.class_name[g] & 1))
      v.GetGresMap()[gres_name(L.g.class_name[g])].specified[gres_type(g)] = t.class_count[g];
  return v;
}
Ctld::MetaResource meta_of(const Layout& L, const cns_usage* u) {   // a USAGE record: entries for the non-zero counts
  Ctld::MetaResource m;
  if (!u) return m;
  m.resource.SetCpuCount(cpu_t::from_raw_value(u->cpu_raw));
  m.resource.SetMemoryBytes(u->mem);
  for (uint32_t n = 0; n < CNS_MAX_GRES_NAMES; ++n)
    if (u->name_total[n]) m.resource.GetGresMap()[gres_name(n)].total = u->name_total[n];
  for (uint32_t g = 0; g < L.g.num_classes; ++g)
    if (u->class_count[g]) m.resource.GetGresMap()[gres_name(L.g.class_name[g])].specified[gres_type(g)] = u->class_count[g];
  m.jobs_count = u->jobs_count;
  m.wall_time = absl::Seconds(u->wall_sec);
  return m;
}
void put_usage(cns_usage* out, uint8_t* ex, size_t i, const Ctld::MetaResource* m) {
  if (ex) ex[i] = m ? 1 : 0;
  if (!out) return;
  cns_usage u{};
  if (m) {
    u.cpu_raw = m->resource.GetCpuCount().raw_value();
    u.mem = m->resource.GetMemoryBytes();
    u.wall_sec = absl::ToInt64Seconds(m->wall_time);
    u.jobs_count = m->jobs_count;
    for (const auto& [name, gc] : m->resource.GetGresMap()) {
      u.name_total[idx_of(name)] = gc.total;
      for (const auto& [type, cnt] : gc.specified) u.class_count[idx_of(type)] = cnt;
    }
  }
  out[i] = u;
}
}  // namespace

int ora_run_limits(const cns_gres_layout* gl, const cns_limit_tables* t, const cns_limit_job_soa* jobs,
                   const cns_placement_soa* pl, uint8_t* reason_out, uint64_t* num_admitted, cns_usage* uq, uint8_t* uqe,
                   cns_usage* up, uint8_t* upe, cns_usage* aq, uint8_t* aqe, cns_usage* ap, uint8_t* ape, cns_usage* qu) {
  using namespace Ctld;
  try {
    Layout L; L.g = *gl;
    const uint32_t Q = t->num_qos, Pn = t->num_partitions;
    auto ex = [](const uint8_t* e, size_t i) { return !e || e[i]; };
    auto us = [&](const cns_usage* u, size_t i) { return meta_of(L, u ? u + i : nullptr); };
    for (uint64_t i = 0; i < jobs->num_jobs; ++i)
      if (jobs->user[i] >= t->num_users || jobs->user_acct[i] >= t->num_user_accts || jobs->account[i] >= t->num_accounts ||
          jobs->qos[i] >= Q || jobs->partition[i] >= Pn)
        return -1;
    // ---- g_account_manager: QoS, accounts (parents, partition limits), users (per-account partition limits) ----
    AccountManager mgr;
    for (uint32_t q = 0; q < Q; ++q) {
      const cns_qos_limits& s = t->qos[q];
      auto qos = std::make_unique<Qos>();
      qos->max_jobs_per_user = s.max_jobs_per_user; qos->max_jobs_per_account = s.max_jobs_per_account; qos->max_jobs = s.max_jobs;
      qos->max_cpus_per_user = cpu_t::from_raw_value(s.max_cpus_per_user_raw);
      qos->max_wall = absl::Seconds(s.max_wall_sec);
      qos->max_tres = limit_view(L, s.max_tres);
      qos->max_tres_per_user = limit_view(L, s.max_tres_per_user);
      qos->max_tres_per_account = limit_view(L, s.max_tres_per_account);
      mgr.qos_map[qos_name(q)] = std::move(qos);
    }
    auto part_limit = [&](uint32_t idx) {
      PartitionResourceLimit pl_;
      pl_.max_tres = limit_view(L, t->part_limits[idx].max_tres);
      pl_.max_jobs = t->part_limits[idx].max_jobs;
      pl_.max_wall = absl::Seconds(t->part_limits[idx].max_wall_sec);
      return pl_;
    };
    for (uint32_t a = 0; a < t->num_accounts; ++a) {
      auto acc = std::make_unique<Account>();
      acc->name = acct_name(a);
      if (t->acct_parent[a] != CNS_LIM_NONE) acc->parent_account = acct_name(t->acct_parent[a]);
      if (t->acct_part_limit)
        for (uint32_t p = 0; p < Pn; ++p)
          if (t->acct_part_limit[(size_t)a * Pn + p] != CNS_LIM_NONE)
            acc->partition_to_limit_map.emplace(part_name(p), part_limit(t->acct_part_limit[(size_t)a * Pn + p]));
      mgr.account_map.emplace(acct_name(a), std::move(acc));
    }
    for (uint32_t u = 0; u < t->num_users; ++u) {
      auto user = std::make_unique<User>();
      user->name = user_name(u);
      mgr.user_map.emplace(user_name(u), std::move(user));
    }
    // the (user, account) pair of a user_acct index arrives with the jobs
    std::vector<uint32_t> user_of_ua(t->num_user_accts, CNS_LIM_NONE), acct_of_ua(t->num_user_accts, CNS_LIM_NONE);
    AccountMetaContainer amc;
    for (uint32_t u = 0; u < t->num_users; ++u) {
      MetaResourceStat& st = amc.m_user_meta_map_[user_name(u)];
      for (uint32_t q = 0; q < Q; ++q)
        if (ex(t->user_qos_exists, (size_t)u * Q + q)) st.qos_to_resource_map[qos_name(q)] = us(t->user_qos, (size_t)u * Q + q);
    }
    for (uint64_t i = 0; i < jobs->num_jobs; ++i) {
      const uint32_t ua = jobs->user_acct[i];
      if (user_of_ua[ua] != CNS_LIM_NONE) continue;
      user_of_ua[ua] = jobs->user[i]; acct_of_ua[ua] = jobs->account[i];
      User& user = *mgr.user_map.at(user_name(jobs->user[i]));
      User::AttrsInAccount& attrs = user.account_to_attrs_map[acct_name(jobs->account[i])];
      MetaResourceStat& st = amc.m_user_meta_map_[user_name(jobs->user[i])];
      for (uint32_t p = 0; p < Pn; ++p) {
        if (t->user_part_limit && t->user_part_limit[(size_t)ua * Pn + p] != CNS_LIM_NONE)
          attrs.partition_to_limit_map.emplace(part_name(p), part_limit(t->user_part_limit[(size_t)ua * Pn + p]));
        if (ex(t->user_part_exists, (size_t)ua * Pn + p))
          st.account_to_partition_to_resource_map[acct_name(jobs->account[i])][part_name(p)] = us(t->user_part, (size_t)ua * Pn + p);
      }
    }
    for (uint32_t a = 0; a < t->num_accounts; ++a) {
      MetaResourceStat& st = amc.m_account_meta_map_[acct_name(a)];
      for (uint32_t q = 0; q < Q; ++q)
        if (ex(t->acct_qos_exists, (size_t)a * Q + q)) st.qos_to_resource_map[qos_name(q)] = us(t->acct_qos, (size_t)a * Q + q);
      for (uint32_t p = 0; p < Pn; ++p)
        if (ex(t->acct_part_exists, (size_t)a * Pn + p)) st.partition_to_resource_map[part_name(p)] = us(t->acct_part, (size_t)a * Pn + p);
    }
    for (uint32_t q = 0; q < Q; ++q) amc.m_qos_meta_map_[qos_name(q)] = us(t->qos_usage, q);
    g_account_manager = &mgr;

    // ---- the commit loop, pending-vector order (JobScheduler.cpp:1492) ---------------------------------
    uint64_t adm = 0;
    for (uint64_t i = 0; i < jobs->num_jobs; ++i) {
      const uint64_t s = jobs->select_index ? jobs->select_index[i] : i;
      if (pl->reason[s] != CNS_REASON_NONE || (jobs->skip && jobs->skip[i])) {   // :1507-1510 and the `continue`s before :1565
        reason_out[i] = CNS_LIM_NOT_CANDIDATE;
        continue;
      }
      JobInCtld j;
      j.job_id = (job_id_t)i;
      j.username = user_name(jobs->user[i]);
      j.account = acct_name(jobs->account[i]);
      for (uint32_t a = jobs->account[i]; a != CNS_LIM_NONE; a = t->acct_parent[a]) j.account_chain.push_back(acct_name(a));
      j.qos = qos_name(jobs->qos[i]);
      j.partition_id = part_name(jobs->partition[i]);
      j.time_limit = absl::Seconds(jobs->time_limit_sec[i]);
      PdJobInScheduler pd(&j);
      for (uint64_t r = pl->place_offsets[s]; r < pl->place_offsets[s + 1]; ++r) {
        if (pl->node_idx[r] == CNS_NODE_NONE) continue;
        MaskRes m;
        m.cpu = pl->cpu_raw[r]; m.mem = pl->mem[r]; m.clo = pl->core_lo[r]; m.chi = pl->core_hi[r]; m.gres = pl->gres[r];
        m.c2 = pl->core_w2 ? pl->core_w2[r] : 0; m.c3 = pl->core_w3 ? pl->core_w3[r] : 0;
        pd.allocated_res.AddResourceInNode(node_name(pl->node_idx[r]), to_ref(L, m));
      }
      const std::expected<void, std::string> result = amc.CheckAndMallocMetaResource(pd);   // AccountMetaContainer.cpp:180-224
      int code = 0;
      if (!result) {
        code = -1;
        for (int c = 1; c < 16; ++c)
          if (result.error() == kLimitReasonStr[c]) { code = c; break; }
        if (code < 0) throw std::runtime_error("reason outside include/crane_gpu/run_limits.h: " + result.error());
      }
      reason_out[i] = (uint8_t)code;
      adm += code == 0;
    }
    if (num_admitted) *num_admitted = adm;

    // ---- usage after the pass, shapes as in cns_limit_tables -------------------------------------------
    auto find2 = [](const auto& map, const std::string& k) -> const MetaResource* {
      auto it = map.find(k);
      return it == map.end() ? nullptr : &it->second;
    };
    for (uint32_t u = 0; u < t->num_users; ++u) {
      const MetaResourceStat& st = amc.m_user_meta_map_.find(user_name(u))->second;
      for (uint32_t q = 0; q < Q; ++q) put_usage(uq, uqe, (size_t)u * Q + q, find2(st.qos_to_resource_map, qos_name(q)));
    }
    for (uint32_t x = 0; x < t->num_user_accts; ++x)
      for (uint32_t p = 0; p < Pn; ++p) {
        const MetaResource* m = nullptr;
        MetaResource tmp;
        if (user_of_ua[x] != CNS_LIM_NONE) {
          const MetaResourceStat& st = amc.m_user_meta_map_.find(user_name(user_of_ua[x]))->second;
          auto a = st.account_to_partition_to_resource_map.find(acct_name(acct_of_ua[x]));
          if (a != st.account_to_partition_to_resource_map.end()) m = find2(a->second, part_name(p));
        } else if (ex(t->user_part_exists, (size_t)x * Pn + p)) {
          tmp = us(t->user_part, (size_t)x * Pn + p);
          m = &tmp;
        }
        put_usage(up, upe, (size_t)x * Pn + p, m);
      }
    for (uint32_t a = 0; a < t->num_accounts; ++a) {
      const MetaResourceStat& st = amc.m_account_meta_map_.find(acct_name(a))->second;
      for (uint32_t q = 0; q < Q; ++q) put_usage(aq, aqe, (size_t)a * Q + q, find2(st.qos_to_resource_map, qos_name(q)));
      for (uint32_t p = 0; p < Pn; ++p) put_usage(ap, ape, (size_t)a * Pn + p, find2(st.partition_to_resource_map, part_name(p)));
    }
    for (uint32_t q = 0; q < Q; ++q) put_usage(qu, nullptr, q, &amc.m_qos_meta_map_.find(qos_name(q))->second);
    g_account_manager = nullptr;
    return 0;
  } catch (const std::exception& e) { g_last_error = e.what(); Ctld::g_account_manager = nullptr; return -2; }
}
const char* ref_limit_reason_string(int code) { return code >= 0 && code < 16 ? kLimitReasonStr[code] : "?"; }

// ---- JobInCtld::SchedulePendingSteps (CtldPublicDefs.cpp:2038-2159), structs of include/crane_gpu/steps.h ----------
// The reference walks step_res_avail_.EachNodeResMap(), an unordered_map.  In the canonical flavour that map is ordered
// by the craned id string; the ids here are zero-padded, so the walk is in ascending dense node index — the order
// include/crane_gpu/steps.h defines.  (libcrane_ref_hash.so walks in libstdc++'s hash order: only for cases where
// the order cannot matter.)
int ora_schedule_steps(const cns_gres_layout* gl, const cns_step_job_soa* jb, const cns_step_soa* st, cns_step_result_soa* out, int) {
  using namespace Ctld;
  try {
    Layout L; L.g = *gl;
    auto nname = [](uint32_t n) { char b[16]; snprintf(b, sizeof b, "n%09u", n); return std::string(b); };
    uint64_t po = 0, to = 0;
    for (uint32_t s = 0; s < st->num_steps; ++s) {
      out->place_offsets[s] = po; out->task_offsets[s] = to;
      po += st->node_num[s]; to += st->ntasks[s];
    }
    out->place_offsets[st->num_steps] = po; out->task_offsets[st->num_steps] = to;
    for (uint64_t i = 0; i < po; ++i) {
      out->node_idx[i] = CNS_NODE_NONE; out->node_ntasks[i] = 0; out->node_cpu_raw[i] = 0; out->node_mem[i] = 0;
      out->node_core_lo[i] = 0; out->node_core_hi[i] = 0; out->node_gres[i] = 0;
      if (out->node_core_w2) out->node_core_w2[i] = 0;
      if (out->node_core_w3) out->node_core_w3[i] = 0;
    }
    for (uint64_t i = 0; i < to; ++i) {
      out->task_node[i] = CNS_NODE_NONE; out->task_cpu_raw[i] = 0; out->task_mem[i] = 0; out->task_core_lo[i] = 0;
      out->task_core_hi[i] = 0; out->task_gres[i] = 0;
      if (out->task_core_w2) out->task_core_w2[i] = 0;
      if (out->task_core_w3) out->task_core_w3[i] = 0;
    }
    for (uint32_t s = 0; s < st->num_steps; ++s) out->scheduled[s] = 0;
    for (uint32_t jx = 0; jx < jb->num_jobs; ++jx) {
      JobInCtld job;
      job.job_id = jx;
      for (uint32_t n = jb->node_offsets[jx]; n < jb->node_offsets[jx + 1]; ++n) {
        MaskRes m;
        m.cpu = jb->avail_cpu_raw[n]; m.mem = jb->avail_mem[n]; m.clo = jb->avail_core_lo[n];
        m.chi = jb->avail_core_hi ? jb->avail_core_hi[n] : 0; m.gres = jb->avail_gres ? jb->avail_gres[n] : 0;
        m.c2 = jb->avail_core_w2 ? jb->avail_core_w2[n] : 0; m.c3 = jb->avail_core_w3 ? jb->avail_core_w3[n] : 0;
        job.step_res_avail_.EachNodeResMap()[nname(jb->node_idx[n])] = to_ref(L, m);   // SetStepResAvail, CtldPublicDefs.h:1140
      }
      std::vector<std::unique_ptr<CommonStepInCtld>> steps;
      for (uint32_t s = jb->step_offsets[jx]; s < jb->step_offsets[jx + 1]; ++s) {
        auto sp = std::make_unique<CommonStepInCtld>();
        sp->job_id = jx; sp->step_id = s;
        sp->req_node_res_view = view_of(L, st->node_cpu_raw ? st->node_cpu_raw[s] : 0, st->node_mem ? st->node_mem[s] : 0,
                                        st->node_gres_total ? st->node_gres_total + (size_t)s * CNS_MAX_GRES_NAMES : nullptr,
                                        st->node_gres_spec ? st->node_gres_spec + (size_t)s * CNS_MAX_GRES_CLASSES : nullptr);
        sp->req_task_res_view = view_of(L, st->task_cpu_raw[s], st->task_mem[s],
                                        st->task_gres_total ? st->task_gres_total + (size_t)s * CNS_MAX_GRES_NAMES : nullptr,
                                        st->task_gres_spec ? st->task_gres_spec + (size_t)s * CNS_MAX_GRES_CLASSES : nullptr);
        sp->node_num = st->node_num[s]; sp->ntasks = st->ntasks[s];
        sp->ntasks_per_node_min = st->ntasks_per_node_min[s]; sp->ntasks_per_node_max = st->ntasks_per_node_max[s];
        if (st->incl_offsets) for (uint32_t k = st->incl_offsets[s]; k < st->incl_offsets[s + 1]; ++k) sp->included_nodes.insert(nname(st->incl_nodes[k]));
        if (st->excl_offsets) for (uint32_t k = st->excl_offsets[s]; k < st->excl_offsets[s + 1]; ++k) sp->excluded_nodes.insert(nname(st->excl_nodes[k]));
        job.m_steps_[s] = sp.get();
        job.pending_step_ids_.push(s);   // AddStep, CtldPublicDefs.h:1108-1113
        steps.push_back(std::move(sp));
      }
      std::vector<CommonStepInCtld*> scheduled;
      job.SchedulePendingSteps(&scheduled);
      for (CommonStepInCtld* sp : scheduled) {
        const uint32_t s = sp->step_id;
        out->scheduled[s] = 1;
        // pop order of the candidate queue = order of the first task id handed to each node (:2109-2128)
        std::map<task_id_t, std::string> first_task;
        for (const auto& [cid, tasks] : sp->craned_task_map) first_task[*tasks.begin()] = cid;
        uint64_t p = out->place_offsets[s];
        for (const auto& [tid, cid] : first_task) {
          const MaskRes m = from_ref(sp->allocated_res.At(cid));
          out->node_idx[p] = (uint32_t)strtoul(cid.c_str() + 1, nullptr, 10);
          out->node_ntasks[p] = (uint32_t)sp->craned_task_map.at(cid).size();
          out->node_cpu_raw[p] = m.cpu; out->node_mem[p] = m.mem; out->node_core_lo[p] = m.clo; out->node_core_hi[p] = m.chi; out->node_gres[p] = m.gres;
          if (out->node_core_w2) out->node_core_w2[p] = m.c2;
          if (out->node_core_w3) out->node_core_w3[p] = m.c3;
          ++p;
        }
        for (const auto& [cid, tasks] : sp->craned_task_map)
          for (task_id_t tid : tasks) {
            const uint64_t q = out->task_offsets[s] + tid;
            const MaskRes m = from_ref(sp->task_res_map.at(tid));
            out->task_node[q] = (uint32_t)strtoul(cid.c_str() + 1, nullptr, 10);
            out->task_cpu_raw[q] = m.cpu; out->task_mem[q] = m.mem; out->task_core_lo[q] = m.clo; out->task_core_hi[q] = m.chi; out->task_gres[q] = m.gres;
            if (out->task_core_w2) out->task_core_w2[q] = m.c2;
            if (out->task_core_w3) out->task_core_w3[q] = m.c3;
          }
      }
      for (uint32_t n = jb->node_offsets[jx]; n < jb->node_offsets[jx + 1]; ++n) {
        const MaskRes m = from_ref(job.step_res_avail_.At(nname(jb->node_idx[n])));
        out->avail_cpu_raw[n] = m.cpu; out->avail_mem[n] = m.mem; out->avail_core_lo[n] = m.clo; out->avail_core_hi[n] = m.chi; out->avail_gres[n] = m.gres;
        if (out->avail_core_w2) out->avail_core_w2[n] = m.c2;
        if (out->avail_core_w3) out->avail_core_w3[n] = m.c3;
      }
    }
    return 0;
  } catch (const crane_ref::RefAssertion& e) { g_last_error = std::string("reference assertion failed: ") + e.what(); return -3;
  } catch (const std::exception& e) { g_last_error = e.what(); return -2; }
}

// ---- LicenseManager::CheckLicenseCountSufficient (LicenseManager.cpp:167-221), the pre-pass NodeSelect runs between ordering and
// selection (JobScheduler.cpp:6739), on flat arrays.  License l = 0..L-1 is named "lic<l>" (an index >= L in a request: a license the
// table does not know); job j asks for entries [req_off[j], req_off[j + 1]) of (req_lic, req_cnt) IN REQUEST ORDER; is_or[j]: the first
// alternative that fits instead of all of them.  Out: rejected[j] = the job left with reason "License"; its actual_licenses as
// (license index, count) sorted by index at [act_off[j], act_off[j + 1]).
int ref_license_check(uint32_t L, const uint32_t* total, const uint32_t* used, const uint32_t* reserved, const uint32_t* last_deficit,
                      uint32_t J, const uint32_t* req_off, const uint32_t* req_lic, const uint32_t* req_cnt, const uint8_t* is_or,
                      uint8_t* rejected, uint32_t* act_off, uint32_t* act_lic, uint32_t* act_cnt) {
  using namespace Ctld;
  try {
    LicenseManager lm;
    for (uint32_t l = 0; l < L; ++l) {
      License lic{};
      lic.license_id = "lic" + std::to_string(l);
      lic.total = total[l]; lic.used = used[l]; lic.reserved = reserved[l]; lic.last_deficit = last_deficit[l];
      lic.remote = false; lic.last_consumed = 0;
      lm.m_licenses_map_.map[lic.license_id].value = lic;
    }
    JobArena<PdJobInScheduler> arena(J);
    std::vector<PdJobInScheduler*> vec;
    for (uint32_t j = 0; j < J; ++j) {
      JobInCtld jc;
      jc.job_id = j;
      PdJobInScheduler* p = arena.make(&jc);
      for (uint32_t x = req_off[j]; x < req_off[j + 1]; ++x) {
        auto* e = p->req_licenses.Add();
        e->key_ = "lic" + std::to_string(req_lic[x]);
        e->count_ = req_cnt[x];
      }
      p->is_license_or = is_or[j] != 0;
      p->actual_licenses.emplace("stale", 1u);   // (the pass must clear what an earlier cycle left: :181)
      vec.push_back(p);
    }
    lm.CheckLicenseCountSufficient(&vec);
    act_off[0] = 0;
    for (uint32_t j = 0; j < J; ++j) {
      const PdJobInScheduler* p = vec[j];
      rejected[j] = p->reason == "License" ? 1 : 0;
      std::vector<std::pair<uint32_t, uint32_t>> a;
      if (req_off[j + 1] != req_off[j])   // (a job without requests is not looked at: its map stays as it was)
        for (const auto& [id, cnt] : p->actual_licenses) a.emplace_back((uint32_t)strtoul(id.c_str() + 3, nullptr, 10), cnt);
      std::sort(a.begin(), a.end());
      uint32_t o = act_off[j];
      for (const auto& [l, c] : a) { act_lic[o] = l; act_cnt[o] = c; ++o; }
      act_off[j + 1] = o;
    }
    return 0;
  } catch (const crane_ref::RefAssertion& e) { g_last_error = std::string("reference assertion failed: ") + e.what(); return -3;
  } catch (const std::exception& e) { g_last_error = e.what(); return -2; }
}

}  // extern "C"
