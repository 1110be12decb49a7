// ORACLE — TEST INFRASTRUCTURE ONLY (see res_algebra.hpp). C entry points over the
// templated restatement so that tests/, smoke() and bench.py's cpu_baseline leg can drive
// it through ctypes with the very same SoA structs as the engine's C ABI
// (include/crane_gpu/node_select.h).  Build: oracle/Makefile  ->  oracle/liboracle.so
#include <chrono>
#include <cstring>
#include <string>

#include "sched_oracle.hpp"
#include "prio_oracle.hpp"
#include "limits_oracle.hpp"
#include "steps_oracle.hpp"
#include "../include/crane_gpu/steps.h"
#include "../include/crane_gpu/preempt.h"

using namespace ora;

namespace {

GresLayout layout_from(const cns_gres_layout& g) {
  GresLayout L;
  L.num_classes = g.num_classes;
  for (u32 i = 0; i < CNS_MAX_GRES_CLASSES; ++i) {
    L.class_name[i] = g.class_name[i];
    L.class_shift[i] = g.class_shift[i];
    L.class_width[i] = g.class_width[i];
  }
  return L;
}

struct OracleRun {
  GresLayout layout;
  std::unique_ptr<SchedOracle<MaskAlgebra>> mask;
  std::unique_ptr<SchedOracle<LitAlgebra>> lit;
  std::vector<std::vector<u32>> part_nodes;
  double seconds = 0;
  u64 jobs_ordered = 0;
};

template <class O>
void emit(O& orc, std::vector<PdJob>& jobs, cns_placement_soa* out) {
  u64 off = 0;
  for (size_t j = 0; j < jobs.size(); ++j) {
    const PdJob& job = jobs[j];
    out->place_offsets[j] = off;
    out->start_sec[j] = job.start_time;
    out->reason[j] = (uint8_t)job.reason;
    u64 k = 0;
    for (const auto& [nid, r] : job.allocated_res) {  // std::map: ascending node index
      u64 q = off + k++;
      out->node_idx[q] = nid;
      out->ntasks[q] = job.craned_id_to_task_num.at(nid);
      out->cpu_raw[q] = r.cpu;
      out->mem[q] = r.mem;
      out->core_lo[q] = r.clo;
      out->core_hi[q] = r.chi;
      if (out->core_w2) out->core_w2[q] = r.c2;
      if (out->core_w3) out->core_w3[q] = r.c3;
      out->gres[q] = r.gres;
    }
    for (; k < job.node_num; ++k) {
      u64 q = off + k;
      out->node_idx[q] = CNS_NODE_NONE;
      out->ntasks[q] = 0;
      out->cpu_raw[q] = 0;
      out->mem[q] = 0;
      out->core_lo[q] = out->core_hi[q] = out->gres[q] = 0;
      if (out->core_w2) out->core_w2[q] = 0;
      if (out->core_w3) out->core_w3[q] = 0;
    }
    off += job.node_num;
  }
  out->place_offsets[jobs.size()] = off;
  (void)orc;
}

}  // namespace

extern "C" {

struct ora_res { int64_t cpu; uint64_t mem, clo, chi, gres, c2, c3; };
struct ora_req { int64_t cpu; uint64_t mem; uint8_t gtot[CNS_MAX_GRES_NAMES]; uint8_t gspec[CNS_MAX_GRES_CLASSES]; };

static MaskRes to_m(const ora_res& r) { MaskRes m; m.cpu = r.cpu; m.mem = r.mem; m.clo = r.clo; m.chi = r.chi; m.gres = r.gres; m.c2 = r.c2; m.c3 = r.c3; return m; }
static ora_res from_m(const MaskRes& m) { return ora_res{m.cpu, m.mem, m.clo, m.chi, m.gres, m.c2, m.c3}; }
static ReqView to_v(const ora_req& q) {
  ReqView v; v.cpu = q.cpu; v.mem = q.mem;
  memcpy(v.gtot, q.gtot, sizeof v.gtot); memcpy(v.gspec, q.gspec, sizeof v.gspec);
  return v;
}

// ---- resource-algebra primitives (algebra: 0 = mask, 1 = literal containers) -------------
int ora_feasible(const cns_gres_layout* gl, int algebra, const ora_req* req, const ora_res* avail, ora_res* out) {
  GresLayout L = layout_from(*gl);
  if (algebra == 0) {
    MaskAlgebra A(&L); MaskRes o;
    bool ok = A.feasible(to_v(*req), to_m(*avail), &o);
    if (ok) *out = from_m(o);
    return ok;
  }
  LitAlgebra A(&L); LitRes o;
  bool ok = A.feasible(to_v(*req), A.from_mask(to_m(*avail)), &o);
  if (ok) *out = from_m(A.to_mask(o));
  return ok;
}
int ora_binop(const cns_gres_layout* gl, int algebra, int op, const ora_res* a, const ora_res* b, ora_res* out) {
  // op: 0 ckmin, 1 add, 2 sub, 3 le (returns bool)
  GresLayout L = layout_from(*gl);
  auto run = [&](auto A) -> int {
    auto x = A.from_mask(to_m(*a));
    auto y = A.from_mask(to_m(*b));
    int ret = 0;
    switch (op) {
      case 0: A.ckmin(x, y); break;
      case 1: A.add(x, y); break;
      case 2: A.sub(x, y); break;
      case 3: ret = A.le(x, y); break;
    }
    if (out) *out = from_m(A.to_mask(x));
    return ret;
  };
  return algebra == 0 ? run(MaskAlgebra(&L)) : run(LitAlgebra(&L));
}

// ---- one scheduling cycle ---------------------------------------------------------------
// Returns an opaque run handle through *run_out (free with ora_free) for the debug getters.
static int ora_select_impl(const cns_config* cfg, const cns_node_soa* nodes, const cns_running_soa* running,
                           const cns_resv_soa* resv, int64_t now, const cns_job_soa* jobs, cns_placement_soa* out,
                           int algebra, void** run_out, const cns_preempt_soa* pre = nullptr, cns_preempt_out* pout = nullptr);
// ... with preemption (include/crane_gpu/preempt.h)
int ora_select_preempt(const cns_config* cfg, const cns_node_soa* nodes, const cns_running_soa* running,
                       const cns_resv_soa* resv, int64_t now, const cns_job_soa* jobs, const cns_preempt_soa* pre,
                       cns_placement_soa* out, cns_preempt_out* pout, int algebra, void** run_out) {
  return ora_select_impl(cfg, nodes, running, resv, now, jobs, out, algebra, run_out, pre, pout);
}
int ora_select(const cns_config* cfg, const cns_node_soa* nodes, const cns_running_soa* running,
               int64_t now, const cns_job_soa* jobs, cns_placement_soa* out, int algebra,
               void** run_out) {
  return ora_select_impl(cfg, nodes, running, nullptr, now, jobs, out, algebra, run_out);
}
// ... with reservations (cns_resv_soa, include/crane_gpu/node_select.h)
int ora_select_resv(const cns_config* cfg, const cns_node_soa* nodes, const cns_running_soa* running,
                    const cns_resv_soa* resv, int64_t now, const cns_job_soa* jobs, cns_placement_soa* out,
                    int algebra, void** run_out) {
  return ora_select_impl(cfg, nodes, running, resv, now, jobs, out, algebra, run_out);
}
static int ora_select_impl(const cns_config* cfg, const cns_node_soa* nodes, const cns_running_soa* running,
                           const cns_resv_soa* resv, int64_t now, const cns_job_soa* jobs, cns_placement_soa* out,
                           int algebra, void** run_out, const cns_preempt_soa* pre, cns_preempt_out* pout) {
  auto run = std::make_unique<OracleRun>();
  run->layout = layout_from(nodes->gres);
  u32 maxjobs = cfg && cfg->max_job_num_per_node ? cfg->max_job_num_per_node : 1000;
  i64 window = cfg && cfg->max_time_window_sec ? cfg->max_time_window_sec : 7 * 24 * 3600;
  u64 batch = cfg ? cfg->scheduled_batch_size : 0;

  std::vector<MaskRes> total(nodes->num_nodes);
  std::vector<uint8_t> sched(nodes->num_nodes, 1);
  for (u32 n = 0; n < nodes->num_nodes; ++n) {
    total[n].cpu = nodes->cpu_total_raw[n];
    total[n].mem = nodes->mem_total[n];
    total[n].clo = nodes->core_lo ? nodes->core_lo[n] : 0;
    total[n].chi = nodes->core_hi ? nodes->core_hi[n] : 0;
    total[n].c2 = nodes->core_w2 ? nodes->core_w2[n] : 0;
    total[n].c3 = nodes->core_w3 ? nodes->core_w3[n] : 0;
    total[n].gres = nodes->gres_slots ? nodes->gres_slots[n] : 0;
    if (nodes->schedulable) sched[n] = nodes->schedulable[n];
  }
  run->part_nodes.resize(nodes->num_partitions);
  for (u32 p = 0; p < nodes->num_partitions; ++p)
    for (u32 i = nodes->part_offsets[p]; i < nodes->part_offsets[p + 1]; ++i)
      run->part_nodes[p].push_back(nodes->part_nodes[i]);

  std::vector<RnJob> rn;
  if (running) {
    rn.resize(running->num_jobs);
    for (u32 r = 0; r < running->num_jobs; ++r) {
      rn[r].end_time = running->end_sec[r];
      for (u32 a = running->alloc_offsets[r]; a < running->alloc_offsets[r + 1]; ++a) {
        MaskRes m;
        m.cpu = running->alloc_cpu_raw[a];
        m.mem = running->alloc_mem[a];
        m.clo = running->alloc_core_lo ? running->alloc_core_lo[a] : 0;
        m.chi = running->alloc_core_hi ? running->alloc_core_hi[a] : 0;
        m.c2 = running->alloc_core_w2 ? running->alloc_core_w2[a] : 0;
        m.c3 = running->alloc_core_w3 ? running->alloc_core_w3[a] : 0;
        m.gres = running->alloc_gres ? running->alloc_gres[a] : 0;
        rn[r].allocs.push_back({running->alloc_node[a], m});
      }
      if (running->reservation) rn[r].reservation = running->reservation[r];
    }
  }
  std::vector<Resv> rv;
  if (resv) {
    rv.resize(resv->num_resv);
    for (u32 v = 0; v < resv->num_resv; ++v) {
      rv[v].start_time = resv->start_sec[v];
      rv[v].end_time = resv->end_sec[v];
      for (u32 a = resv->alloc_offsets[v]; a < resv->alloc_offsets[v + 1]; ++a) {
        MaskRes m;
        m.cpu = resv->alloc_cpu_raw[a];
        m.mem = resv->alloc_mem[a];
        m.clo = resv->alloc_core_lo ? resv->alloc_core_lo[a] : 0;
        m.chi = resv->alloc_core_hi ? resv->alloc_core_hi[a] : 0;
        m.c2 = resv->alloc_core_w2 ? resv->alloc_core_w2[a] : 0;
        m.c3 = resv->alloc_core_w3 ? resv->alloc_core_w3[a] : 0;
        m.gres = resv->alloc_gres ? resv->alloc_gres[a] : 0;
        rv[v].allocs.push_back({resv->alloc_node[a], m});
      }
    }
  }

  std::vector<PdJob> pd(jobs->num_jobs);
  for (u64 j = 0; j < jobs->num_jobs; ++j) {
    PdJob& q = pd[j];
    q.partition = jobs->partition[j];
    q.time_limit = jobs->time_limit_sec[j];
    q.req_node.cpu = jobs->node_cpu_raw ? jobs->node_cpu_raw[j] : 0;
    q.req_node.mem = jobs->node_mem[j];
    q.req_task.cpu = jobs->task_cpu_raw[j];
    q.req_task.mem = jobs->task_mem[j];
    if (jobs->gres_total) memcpy(q.req_node.gtot, jobs->gres_total + j * CNS_MAX_GRES_NAMES, CNS_MAX_GRES_NAMES);
    if (jobs->gres_spec) memcpy(q.req_node.gspec, jobs->gres_spec + j * CNS_MAX_GRES_CLASSES, CNS_MAX_GRES_CLASSES);
    q.node_num = jobs->node_num[j];
    q.ntasks = jobs->ntasks[j];
    q.tpn_min = jobs->ntasks_per_node_min[j];
    q.tpn_max = jobs->ntasks_per_node_max[j];
    q.exclusive = jobs->exclusive ? jobs->exclusive[j] != 0 : false;
    if (jobs->incl_offsets)
      for (u64 i = jobs->incl_offsets[j]; i < jobs->incl_offsets[j + 1]; ++i) q.included_nodes.insert(jobs->incl_nodes[i]);
    if (jobs->excl_offsets)
      for (u64 i = jobs->excl_offsets[j]; i < jobs->excl_offsets[j + 1]; ++i) q.excluded_nodes.insert(jobs->excl_nodes[i]);
    q.skip = jobs->skip ? jobs->skip[j] != 0 : false;
    if (jobs->reservation) q.reservation = jobs->reservation[j];
  }
  PreemptCfg pcfg;
  PreemptCfg* pc = nullptr;
  if (pre) {  // include/crane_gpu/preempt.h
    pc = &pcfg;
    pcfg.enabled = pre->enabled != 0;
    pcfg.qos_preempt.resize(pre->num_qos);
    for (u32 q = 0; q < pre->num_qos; ++q)
      for (u32 i = pre->qos_preempt_offsets[q]; i < pre->qos_preempt_offsets[q + 1]; ++i) pcfg.qos_preempt[q].push_back(pre->qos_preempt[i]);
    for (u64 j = 0; j < jobs->num_jobs; ++j) {
      pd[j].job_id = pre->pd_job_id ? pre->pd_job_id[j] : (u32)j;
      pd[j].qos = pre->pd_qos ? pre->pd_qos[j] : 0;
      pd[j].qos_priority = pre->pd_qos_priority ? pre->pd_qos_priority[j] : 0;
      pd[j].priority = pre->pd_priority ? pre->pd_priority[j] : 0.0;
    }
    for (size_t r = 0; r < rn.size(); ++r) {
      rn[r].job_id = pre->rn_job_id ? pre->rn_job_id[r] : (u32)r;
      rn[r].qos = pre->rn_qos ? pre->rn_qos[r] : 0;
      rn[r].qos_priority = pre->rn_qos_priority ? pre->rn_qos_priority[r] : 0;
      rn[r].start_time = pre->rn_start_sec ? pre->rn_start_sec[r] : 0;
    }
    for (u32 i = 0; i < pre->num_preempting; ++i) pcfg.preempting_set.insert(pre->preempting_job_ids[i]);
  }

  auto t0 = std::chrono::steady_clock::now();  // the reference's own bracket, JobScheduler.cpp:1439-1447
  if (algebra == 0) {
    run->mask = std::make_unique<SchedOracle<MaskAlgebra>>(MaskAlgebra(&run->layout), maxjobs, window);
    run->mask->NodeSelect(now, total, sched, run->part_nodes, rn, pd, batch, rv, pc);
    run->jobs_ordered = run->mask->jobs_ordered();
  } else {
    run->lit = std::make_unique<SchedOracle<LitAlgebra>>(LitAlgebra(&run->layout), maxjobs, window);
    run->lit->NodeSelect(now, total, sched, run->part_nodes, rn, pd, batch, rv, pc);
    run->jobs_ordered = run->lit->jobs_ordered();
  }
  run->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

  if (out) {
    if (run->mask) emit(*run->mask, pd, out);
    else emit(*run->lit, pd, out);
  }
  if (pre && pout) {
    u64 off = 0;
    for (size_t j = 0; j < pd.size(); ++j) {
      pout->offsets[j] = off;
      for (const auto& [is_pd, idx] : pd[j].preempted_jobs) {
        if (off >= pout->capacity) return -7;
        pout->preempted[off++] = is_pd ? (idx | CNS_PREEMPT_REF_PENDING) : idx;
      }
    }
    pout->offsets[pd.size()] = off;
    pout->num_cancelled = 0;
    for (u32 id : pcfg.cancelled) { if (pout->num_cancelled >= pout->cancel_capacity) return -7; pout->cancelled_job_ids[pout->num_cancelled++] = id; }
    pout->num_preempting = 0;
    for (u32 id : pcfg.preempting_set) { if (pout->num_preempting >= pout->preempting_capacity) return -7; pout->preempting_job_ids[pout->num_preempting++] = id; }
  }
  if (run_out) *run_out = run.release();
  return 0;
}

double ora_seconds(void* h) { return static_cast<OracleRun*>(h)->seconds; }
uint64_t ora_jobs_ordered(void* h) { return static_cast<OracleRun*>(h)->jobs_ordered; }

int ora_get_costs(void* h, double* cost_by_part_slot) {
  auto* run = static_cast<OracleRun*>(h);
  size_t q = 0;
  for (u32 p = 0; p < run->part_nodes.size(); ++p)
    for (u32 n : run->part_nodes[p]) {
      bool has = run->mask ? run->mask->HasNode(n) : run->lit->HasNode(n);
      cost_by_part_slot[q++] = !has ? 0.0 : (run->mask ? run->mask->CostOf(p, n) : run->lit->CostOf(p, n));
    }
  return 0;
}

int ora_get_timeline(void* h, uint32_t node, uint32_t capacity, uint32_t* len, int64_t* t,
                     int64_t* cpu_raw, uint64_t* mem, uint64_t* core_lo, uint64_t* core_hi,
                     uint64_t* gres) {
  auto* run = static_cast<OracleRun*>(h);
  auto dump = [&](auto& orc) {
    if (!orc.HasNode(node)) { *len = 0; return 0; }
    const auto& tl = orc.Timeline(node);
    *len = (uint32_t)tl.size();
    uint32_t i = 0;
    for (const auto& [time, res] : tl) {
      if (i >= capacity) break;
      MaskRes m = orc.alg().to_mask(res);
      t[i] = time; cpu_raw[i] = m.cpu; mem[i] = m.mem; core_lo[i] = m.clo; core_hi[i] = m.chi; gres[i] = m.gres;
      ++i;
    }
    return 0;
  };
  return run->mask ? dump(*run->mask) : dump(*run->lit);
}

// ... and the core ids 128..255 of the same entries
int ora_get_timeline_cores(void* h, uint32_t node, uint32_t capacity, uint64_t* core_w2, uint64_t* core_w3) {
  auto* run = static_cast<OracleRun*>(h);
  auto dump = [&](auto& orc) {
    if (!orc.HasNode(node)) return 0;
    uint32_t i = 0;
    for (const auto& [time, res] : orc.Timeline(node)) {
      if (i >= capacity) break;
      MaskRes m = orc.alg().to_mask(res);
      core_w2[i] = m.c2; core_w3[i] = m.c3;
      ++i;
    }
    return 0;
  };
  return run->mask ? dump(*run->mask) : dump(*run->lit);
}

void ora_free(void* h) { delete static_cast<OracleRun*>(h); }


// MultiFactorPriority restatement (prio_oracle.hpp).  Arrays as in include/crane_gpu/priority.h.
int ora_priority_order(int64_t now, uint64_t max_age, uint32_t w_age, uint32_t w_fair, uint32_t w_size, uint32_t w_part,
                       uint32_t w_qos, uint32_t favor_small, uint32_t num_accounts, uint32_t J, const int64_t* submit,
                       const uint32_t* qos, const uint32_t* part, const uint32_t* node_num, const int64_t* cpu_raw,
                       const uint64_t* mem, const uint32_t* account, const double* cached, uint32_t R,
                       const int64_t* r_start, const uint32_t* r_qos, const uint32_t* r_part, const uint32_t* r_node_num,
                       const int64_t* r_cpu_raw, const uint64_t* r_mem, const uint32_t* r_account, uint32_t* order_out,
                       double* prio_out) {
  ora::PrioConfig cfg{max_age, w_age, w_fair, w_size, w_part, w_qos, favor_small != 0};
  std::vector<ora::PrioPending> pd(J);
  for (uint32_t i = 0; i < J; ++i)
    pd[i] = ora::PrioPending{submit[i], qos[i], part[i], node_num[i], cpu_raw[i], mem[i], account[i], cached ? cached[i] : 0.0};
  std::vector<ora::PrioRunning> rn(R);
  for (uint32_t i = 0; i < R; ++i)
    rn[i] = ora::PrioRunning{r_start[i], r_qos[i], r_part[i], r_node_num[i], r_cpu_raw[i], r_mem[i], r_account[i]};
  for (uint32_t i = 0; i < J; ++i) if (account[i] >= num_accounts) return -1;
  for (uint32_t i = 0; i < R; ++i) if (r_account[i] >= num_accounts) return -1;
  std::vector<double> prio;
  const std::vector<uint32_t> order = ora::priority_order(now, cfg, num_accounts, pd, rn, prio);
  for (uint32_t i = 0; i < J; ++i) { order_out[i] = order[i]; prio_out[i] = prio[i]; }
  return 0;
}

// Run-limit admission restatement (limits_oracle.hpp).  Structs as in include/crane_gpu/run_limits.h; the
// placements are the ones a NodeSelect run produced (cns_placement_soa of the oracle or of the engine).
// Usage tables after the pass are written to the (optional) out pointers, shapes as in cns_limit_tables.
int ora_run_limits(const cns_gres_layout* gl, const cns_limit_tables* t, const cns_limit_job_soa* jobs,
                   const cns_placement_soa* pl, uint8_t* reason_out, uint64_t* num_admitted, cns_usage* uq, uint8_t* uqe,
                   cns_usage* up, uint8_t* upe, cns_usage* aq, uint8_t* aqe, cns_usage* ap, uint8_t* ape, cns_usage* qu) {
  lim_oracle::Limits L(*t, *gl);
  uint64_t adm = 0;
  for (uint64_t i = 0; i < jobs->num_jobs; ++i) {  // commit loop, pending-vector order (JobScheduler.cpp:1492)
    const uint64_t s = jobs->select_index ? jobs->select_index[i] : i;
    if (pl->reason[s] != CNS_REASON_NONE || (jobs->skip && jobs->skip[i])) {  // :1507-1510 and the `continue`s before :1565
      reason_out[i] = CNS_LIM_NOT_CANDIDATE;
      continue;
    }
    if (jobs->user[i] >= t->num_users || jobs->user_acct[i] >= t->num_user_accts || jobs->account[i] >= t->num_accounts ||
        jobs->qos[i] >= t->num_qos || jobs->partition[i] >= t->num_partitions)
      return -1;
    // job.allocated_res.View() (PublicHeader.cpp:946-952)
    int64_t cpu = 0;
    uint64_t mem = 0, cc[CNS_MAX_GRES_CLASSES] = {0};
    for (uint64_t r = pl->place_offsets[s]; r < pl->place_offsets[s + 1]; ++r) {
      if (pl->node_idx[r] == CNS_NODE_NONE) continue;
      cpu += pl->cpu_raw[r];
      mem += pl->mem[r];
      for (uint32_t g = 0; g < gl->num_classes; ++g) {
        const uint64_t w = gl->class_width[g] >= 64 ? ~0ull : ((1ull << gl->class_width[g]) - 1ull);
        cc[g] += (uint64_t)__builtin_popcountll(pl->gres[r] & (w << gl->class_shift[g]));
      }
    }
    const int r = L.check_and_malloc(jobs->user[i], jobs->user_acct[i], jobs->account[i], jobs->qos[i], jobs->partition[i],
                                     jobs->time_limit_sec[i], L.view_of_counts(cpu, mem, cc));
    reason_out[i] = (uint8_t)r;
    adm += r == 0;
  }
  if (num_admitted) *num_admitted = adm;
  L.export_usage(t->num_users, t->num_user_accts, uq, uqe, up, upe, aq, aqe, ap, ape, qu);
  return 0;
}

}  // extern "C"

// SchedulePendingSteps restatement (steps_oracle.hpp) behind the structs of include/crane_gpu/steps.h.
template <class Alg>
static int run_steps(const Alg& alg, const cns_step_job_soa* jb, const cns_step_soa* st, cns_step_result_soa* out) {
  u64 po = 0, to = 0;
  for (u32 s = 0; s < st->num_steps; ++s) {
    out->place_offsets[s] = po; out->task_offsets[s] = to;
    po += st->node_num[s]; to += st->ntasks[s];
  }
  out->place_offsets[st->num_steps] = po; out->task_offsets[st->num_steps] = to;
  for (u64 i = 0; i < po; ++i) { out->node_idx[i] = CNS_NODE_NONE; out->node_ntasks[i] = 0; out->node_cpu_raw[i] = 0; out->node_mem[i] = 0; out->node_core_lo[i] = 0; out->node_core_hi[i] = 0; out->node_gres[i] = 0;
    if (out->node_core_w2) out->node_core_w2[i] = 0;
    if (out->node_core_w3) out->node_core_w3[i] = 0; }
  for (u64 i = 0; i < to; ++i) { out->task_node[i] = CNS_NODE_NONE; out->task_cpu_raw[i] = 0; out->task_mem[i] = 0; out->task_core_lo[i] = 0; out->task_core_hi[i] = 0; out->task_gres[i] = 0;
    if (out->task_core_w2) out->task_core_w2[i] = 0;
    if (out->task_core_w3) out->task_core_w3[i] = 0; }
  for (u32 j = 0; j < jb->num_jobs; ++j) {
    std::vector<u32> nodes;
    std::vector<typename Alg::Res> avail;
    for (u32 n = jb->node_offsets[j]; n < jb->node_offsets[j + 1]; ++n) {
      nodes.push_back(jb->node_idx[n]);
      MaskRes m;
      m.cpu = jb->avail_cpu_raw[n]; m.mem = jb->avail_mem[n]; m.clo = jb->avail_core_lo[n];
      m.chi = jb->avail_core_hi ? jb->avail_core_hi[n] : 0; m.gres = jb->avail_gres ? jb->avail_gres[n] : 0;
      m.c2 = jb->avail_core_w2 ? jb->avail_core_w2[n] : 0; m.c3 = jb->avail_core_w3 ? jb->avail_core_w3[n] : 0;
      avail.push_back(alg.from_mask(m));
    }
    std::vector<ora::StepReq> steps;
    for (u32 s = jb->step_offsets[j]; s < jb->step_offsets[j + 1]; ++s) {
      ora::StepReq r;
      r.node_view.cpu = st->node_cpu_raw ? st->node_cpu_raw[s] : 0; r.node_view.mem = st->node_mem ? st->node_mem[s] : 0;
      r.task_view.cpu = st->task_cpu_raw[s]; r.task_view.mem = st->task_mem[s];
      for (u32 a = 0; a < CNS_MAX_GRES_NAMES; ++a) {
        r.node_view.gtot[a] = st->node_gres_total ? st->node_gres_total[s * CNS_MAX_GRES_NAMES + a] : 0;
        r.task_view.gtot[a] = st->task_gres_total ? st->task_gres_total[s * CNS_MAX_GRES_NAMES + a] : 0;
      }
      for (u32 g = 0; g < CNS_MAX_GRES_CLASSES; ++g) {
        r.node_view.gspec[g] = st->node_gres_spec ? st->node_gres_spec[s * CNS_MAX_GRES_CLASSES + g] : 0;
        r.task_view.gspec[g] = st->task_gres_spec ? st->task_gres_spec[s * CNS_MAX_GRES_CLASSES + g] : 0;
      }
      r.node_num = st->node_num[s]; r.ntasks = st->ntasks[s]; r.tmin = st->ntasks_per_node_min[s]; r.tmax = st->ntasks_per_node_max[s];
      if (st->incl_offsets) r.included.assign(st->incl_nodes + st->incl_offsets[s], st->incl_nodes + st->incl_offsets[s + 1]);
      if (st->excl_offsets) r.excluded.assign(st->excl_nodes + st->excl_offsets[s], st->excl_nodes + st->excl_offsets[s + 1]);
      steps.push_back(std::move(r));
    }
    const std::vector<ora::StepOut> res = ora::SchedulePendingSteps(alg, nodes, avail, steps);
    for (size_t k = 0; k < res.size(); ++k) {
      const u32 s = jb->step_offsets[j] + (u32)k;
      out->scheduled[s] = res[k].scheduled ? 1 : 0;
      u64 p = out->place_offsets[s], t = out->task_offsets[s];
      for (size_t i = 0; i < res[k].node.size(); ++i, ++p) {
        const MaskRes& m = res[k].node_alloc[i];
        out->node_idx[p] = res[k].node[i]; out->node_ntasks[p] = res[k].node_ntasks[i];
        out->node_cpu_raw[p] = m.cpu; out->node_mem[p] = m.mem; out->node_core_lo[p] = m.clo; out->node_core_hi[p] = m.chi; out->node_gres[p] = m.gres;
        if (out->node_core_w2) out->node_core_w2[p] = m.c2;
        if (out->node_core_w3) out->node_core_w3[p] = m.c3;
      }
      for (size_t i = 0; i < res[k].task_node.size(); ++i, ++t) {
        const MaskRes& m = res[k].task_alloc[i];
        out->task_node[t] = res[k].task_node[i];
        out->task_cpu_raw[t] = m.cpu; out->task_mem[t] = m.mem; out->task_core_lo[t] = m.clo; out->task_core_hi[t] = m.chi; out->task_gres[t] = m.gres;
        if (out->task_core_w2) out->task_core_w2[t] = m.c2;
        if (out->task_core_w3) out->task_core_w3[t] = m.c3;
      }
    }
    for (u32 n = jb->node_offsets[j], i = 0; n < jb->node_offsets[j + 1]; ++n, ++i) {
      const MaskRes m = alg.to_mask(avail[i]);
      out->avail_cpu_raw[n] = m.cpu; out->avail_mem[n] = m.mem; out->avail_core_lo[n] = m.clo; out->avail_core_hi[n] = m.chi; out->avail_gres[n] = m.gres;
      if (out->avail_core_w2) out->avail_core_w2[n] = m.c2;
      if (out->avail_core_w3) out->avail_core_w3[n] = m.c3;
    }
  }
  return 0;
}

extern "C" int ora_schedule_steps(const cns_gres_layout* gl, const cns_step_job_soa* jobs, const cns_step_soa* steps,
                                  cns_step_result_soa* out, int algebra) {
  const GresLayout L = layout_from(*gl);
  if (algebra == 1) { LitAlgebra alg(&L); return run_steps(alg, jobs, steps, out); }
  MaskAlgebra alg(&L);
  return run_steps(alg, jobs, steps, out);
}
