// ORACLE — TEST INFRASTRUCTURE ONLY (see res_algebra.hpp header).  PINNED: the reference ships no test of
// NodeSelect; this file restates it line by line and tests/test_ref_pin.py holds it to the reference's own
// SchedulerAlgo::NodeSelect compiled from /root/reference (oracle/_ref).
//
// CPU restatement of CraneCtld's node-selection cycle, templated on the resource algebra:
//   SchedulerAlgo::NodeSelect                 src/CraneCtld/JobScheduler.cpp:6507-6836
//   LocalScheduler::CalculateRunningNodesAndStartTime_ / GetNodesAndTrySchedule_ / Backfill_
//                                             src/CraneCtld/JobScheduler.cpp:6127-6376
//   NodeState (InitTimeAvailResMap, UpdateResourceInNode)   src/CraneCtld/JobScheduler.h:272-460
//   NodeSelector / NodeRater                  src/CraneCtld/JobScheduler.h:492-595
//   MinCpuTimeRatioFirst                      src/CraneCtld/JobScheduler.h:41-55
//   EarliestStartSubsetSelector & friends     src/CraneCtld/JobScheduler.h:678-865
//   BasicPriority                             src/CraneCtld/JobScheduler.h:183-201
// It keeps the reference's containers (std::map time maps, std::set cost order,
// std::priority_queue top-k heaps, std::list tracker) so that libstdc++'s tie behaviour is
// inherited, with two canonicalisations: nodes are dense ints (cost ties
// break on the node index instead of a NodeState* address) and per-job node lists are
// reported sorted by node index (the reference's order comes from unordered_map iteration).
//   LocalScheduler::TryPreempt_               src/CraneCtld/JobScheduler.cpp:6378-6505
//   PreemptSegTree                            src/CraneCtld/JobScheduler.h:867-980
//   UpdateNodeSelectorWith{Scheduled,Preempted}Job   src/CraneCtld/JobScheduler.h:630-670
// Out of this slice, as in SURVEY.md §8: licenses.
// Compile with -ffp-contract=off: the fp64 cost must not be FMA-contracted.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <list>
#include <map>
#include <memory>
#include <queue>
#include <set>
#include <unordered_map>
#include <utility>
#include <vector>

#include "res_algebra.hpp"
#include "../include/crane_gpu/preempt.h"

namespace ora {

constexpr i64 kInfiniteFuture = INT64_MAX;

struct PdJob {  // PdJobInScheduler, JobScheduler.h:92-170 (fields used on this path)
  u32 partition = 0;
  i64 time_limit = 0;
  ReqView req_node;
  ReqView req_task;
  u32 node_num = 1, ntasks = 1, tpn_min = 1, tpn_max = 1;
  bool exclusive = false;
  std::set<u32> included_nodes, excluded_nodes;
  bool skip = false;
  u32 reservation = CNS_RESV_NONE;  // job->reservation (empty = CNS_RESV_NONE)
  // preemption (JobScheduler.h:113-134): job id, qos (dense id), qos_priority, priority
  u32 job_id = 0, qos = 0, qos_priority = 0;
  double priority = 0.0;
  // results
  i64 start_time = 0, end_time = 0;
  int reason = CNS_REASON_NONE;
  std::map<u32, MaskRes> allocated_res;          // ResourceV3 (node -> ResourceInNodeV3)
  std::map<u32, u32> craned_id_to_task_num;
  std::vector<u32> craned_ids;
  std::vector<std::pair<bool, u32>> preempted_jobs;  // (is_pending, index): std::variant<PdJobInScheduler*, RnJobInScheduler*>
};

struct RnAlloc { u32 node; MaskRes res; };
struct RnJob {
  i64 end_time; std::vector<RnAlloc> allocs; u32 reservation = CNS_RESV_NONE;
  u32 job_id = 0, qos = 0, qos_priority = 0;   // JobScheduler.h:56-70
  i64 start_time = 0;
};
// g_config.Preempt + what NodeSelect reads of the QoS table and keeps across cycles (JobScheduler.cpp:6522-6559, h:984)
struct PreemptCfg {
  bool enabled = false;                          // PreemptType != PREEMPT_NONE (PREEMPT_QOS is the only other mode)
  std::vector<std::vector<u32>> qos_preempt;     // qos id -> the qos ids it may preempt (Qos::preempt)
  std::set<u32> preempting_set;                  // m_preempting_set_: running job ids being preempted (in / out)
  std::vector<u32> cancelled;                    // out: EnqueuePreemptCancel, in call order
};
// ResvMeta (start_time, end_time, res_total per node) as read at JobScheduler.cpp:6627-6679
struct Resv { i64 start_time, end_time; std::vector<RnAlloc> allocs; };

template <class A>
class SchedOracle {
 public:
  using Res = typename A::Res;
  using TimeAvailResMap = std::map<i64, Res>;  // JobScheduler.h:245

  struct NodeState {  // JobScheduler.h:272-460
    struct AllocatedRes { i64 end_time; Res res; };
    struct ReservedRes { i64 start_time, end_time; Res res; };
    u32 idx;
    Res res_total;
    Res res_avail;
    std::vector<AllocatedRes> allocated_res;
    std::vector<ReservedRes> reserved_res;  // future reservations, JobScheduler.h:288
    TimeAvailResMap time_avail_res_map;
    // JobScheduler.h:293-296: qos -> jobs holding resources on the node.  (is_pending, index) stands for the variant of
    // pointers; the reference's flat_hash_set order is unspecified, here it is (pending first, ascending index).
    std::map<u32, std::set<std::pair<bool, u32>>> qos_job_map;
  };

  SchedOracle(const A& alg, u32 max_job_num_per_node, i64 max_time_window)
      : A_(alg), kAlgoMaxJobNumPerNode(max_job_num_per_node), kAlgoMaxTimeWindow(max_time_window) {}

  // -- NodeState::InitTimeAvailResMap, JobScheduler.h:301-338 ---------------------------
  void InitTimeAvailResMap(NodeState& ns, i64 now, i64 end = kInfiniteFuture) {
    std::vector<std::pair<i64, std::pair<bool, const Res*>>> resource_changes;
    for (auto& rr : ns.reserved_res) {  // :305-308
      resource_changes.emplace_back(rr.start_time, std::make_pair(true, &rr.res));
      resource_changes.emplace_back(rr.end_time, std::make_pair(false, &rr.res));
    }
    for (auto& ar : ns.allocated_res) {
      resource_changes.emplace_back(ar.end_time, std::make_pair(false, &ar.res));
      A_.sub(ns.res_avail, ar.res);
    }
    std::stable_sort(resource_changes.begin(), resource_changes.end(),
                     [](const auto& l, const auto& r) {
                       return l.first < r.first ||
                              (l.first == r.first && l.second.first < r.second.first);
                     });
    auto [cur_iter, ok] = ns.time_avail_res_map.emplace(now, ns.res_avail);
    for (const auto& change : resource_changes) {
      if (change.first != cur_iter->first)
        std::tie(cur_iter, ok) = ns.time_avail_res_map.emplace(change.first, cur_iter->second);
      if (change.second.first) A_.sub(cur_iter->second, *change.second.second);
      else A_.add(cur_iter->second, *change.second.second);
    }
    A_.set_zero(ns.time_avail_res_map[end]);
  }

  // -- NodeState::UpdateResourceInNode, JobScheduler.h:340-459 ----------------------------
  void UpdateResourceInNode(NodeState& ns, i64 start_time, i64 end_time, const Res& res, bool is_release = false) {
    auto& m = ns.time_avail_res_map;
    bool ok;
    auto apply = [&](Res& slot) {
      if (is_release) A_.add(slot, res);
      else { assert(A_.le(res, slot)); A_.sub(slot, res); }
    };
    auto job_duration_begin_it = m.upper_bound(start_time);
    if (job_duration_begin_it == m.end()) {
      --job_duration_begin_it;
      typename TimeAvailResMap::iterator inserted_it;
      std::tie(inserted_it, ok) = m.emplace(end_time, job_duration_begin_it->second);
      assert(ok);
      if (job_duration_begin_it->first == start_time) {  // Case #1
        apply(job_duration_begin_it->second);
      } else {  // Case #2
        std::tie(inserted_it, ok) = m.emplace(start_time, job_duration_begin_it->second);
        assert(ok);
        apply(inserted_it->second);
      }
    } else {
      --job_duration_begin_it;
      if (job_duration_begin_it->first != start_time) {  // Case #3
        typename TimeAvailResMap::iterator inserted_it;
        std::tie(inserted_it, ok) = m.emplace(start_time, job_duration_begin_it->second);
        assert(ok);
        job_duration_begin_it = inserted_it;
      }
      auto job_duration_end_it = std::prev(m.upper_bound(end_time));
      for (auto it = job_duration_begin_it; it != job_duration_end_it; it++) apply(it->second);
      if (job_duration_end_it->first != end_time) {
        typename TimeAvailResMap::iterator inserted_it;
        std::tie(inserted_it, ok) = m.emplace(end_time, job_duration_end_it->second);
        assert(ok);
        apply(job_duration_end_it->second);
      }
    }
  }

  // -- MinCpuTimeRatioFirst::UpdateCost, JobScheduler.h:43-54 --------------------------
  static void UpdateCost(double& cost, i64 start_time, i64 end_time, i64 res_cpu_raw,
                         i64 total_cpu_raw, bool is_release = false) {
    // static_cast<double>(cpu_t) == double(raw) / 256.0 (fpm)
    double a = static_cast<double>(res_cpu_raw) / 256.0;
    double t = static_cast<double>(total_cpu_raw) / 256.0;
    double ratio = a / t;
    double delta = static_cast<double>(end_time - start_time) * ratio;
    if (is_release) cost -= delta;
    else cost += delta;
  }

  // -- NodeSelector, JobScheduler.h:492-595 --------------------------------------------
  struct NodeSelector {
    struct NodeRater { NodeState* node_state; double cost; };
    std::unordered_map<u32, NodeRater> m_node_info_map_;
    std::set<std::pair<double, u32>> m_cost_node_info_set_;  // (cost, dense node index)
  };

  struct LocalScheduler { NodeSelector sel; };

  void AddNode(LocalScheduler& ls, i64 now, NodeState* ns) {  // :540-550 + NodeRater ctor :498-511
    double cost = 0.0;
    for (const auto& rr : ns->reserved_res)  // :502-506
      UpdateCost(cost, rr.start_time, rr.end_time, cpu_of(rr.res), cpu_of(ns->res_total));
    for (const auto& ar : ns->allocated_res)
      UpdateCost(cost, now, ar.end_time, cpu_of(ar.res), cpu_of(ns->res_total));
    ls.sel.m_node_info_map_.emplace(ns->idx, typename NodeSelector::NodeRater{ns, cost});
    ls.sel.m_cost_node_info_set_.emplace(cost, ns->idx);
  }

  void AllocateResource(LocalScheduler& ls, i64 start_time, i64 end_time,
                        const std::map<u32, MaskRes>& res) {  // :567-575, :526-538
    for (const auto& [node, mres] : res) {
      auto& info = ls.sel.m_node_info_map_.at(node);
      Res r = A_.from_mask(mres);
      UpdateResourceInNode(*info.node_state, start_time, end_time, r);
      ls.sel.m_cost_node_info_set_.erase({info.cost, node});
      UpdateCost(info.cost, start_time, end_time, mres.cpu, cpu_of(info.node_state->res_total));
      ls.sel.m_cost_node_info_set_.emplace(info.cost, node);
    }
  }

  void ReleaseResourceIfPresent(LocalScheduler& ls, i64 start_time, i64 end_time,
                                const std::map<u32, MaskRes>& res) {  // :577-587
    for (const auto& [node, mres] : res) {
      auto it = ls.sel.m_node_info_map_.find(node);
      if (it == ls.sel.m_node_info_map_.end()) continue;
      auto& info = it->second;
      Res r = A_.from_mask(mres);
      UpdateResourceInNode(*info.node_state, start_time, end_time, r, /*is_release=*/true);
      ls.sel.m_cost_node_info_set_.erase({info.cost, node});
      UpdateCost(info.cost, start_time, end_time, mres.cpu, cpu_of(info.node_state->res_total), /*is_release=*/true);
      ls.sel.m_cost_node_info_set_.emplace(info.cost, node);
    }
  }

  // -- PreemptSegTree, JobScheduler.h:867-980 --------------------------------------------------
  // Times are absl::Time there; `mid = st + (ed - st) / 2` halves an absl::Duration, whose resolution is a quarter of a
  // nanosecond, truncating.  Here: ticks of 1/4 ns RELATIVE to `now` (the arithmetic is translation-invariant and
  // absolute ticks would overflow int64 in 2042).
  static constexpr i64 kTicksPerSecond = 4000000000ll;
  class PreemptSegTree {
    struct Node {
      i64 st, ed;
      Node* ls;
      Node* rs;
      bool satisfied;
      Res res;
      Res add_tag;
      Res sub_tag;
    };
    void add_res_(Node* node, const Res& res) {
      A_->add(node->res, res);
      node->satisfied = A_->le(m_target_res_, node->res);
      if (node->ls) A_->add(node->add_tag, res);
    }
    void sub_res_(Node* node, const Res& res) {
      A_->sub(node->res, res);
      node->satisfied = A_->le(m_target_res_, node->res);
      if (node->ls) A_->add(node->sub_tag, res);
    }
    Node* make(i64 st, i64 ed, bool satisfied, const Res* res) {
      Node* n = new Node{st, ed, nullptr, nullptr, satisfied, Res{}, Res{}, Res{}};
      A_->set_zero(n->res); A_->set_zero(n->add_tag); A_->set_zero(n->sub_tag);
      if (res) n->res = *res;
      return n;
    }
    void push_down_(Node* node) {
      if (node->ls == nullptr) {
        i64 mid = node->st + (node->ed - node->st) / 2;
        node->ls = make(node->st, mid, node->satisfied, &node->res);
        node->rs = make(mid, node->ed, node->satisfied, &node->res);
        return;
      }
      if (!A_->is_zero(node->add_tag)) {
        add_res_(node->ls, node->add_tag);
        add_res_(node->rs, node->add_tag);
        A_->set_zero(node->add_tag);
      }
      if (!A_->is_zero(node->sub_tag)) {
        sub_res_(node->ls, node->sub_tag);
        sub_res_(node->rs, node->sub_tag);
        A_->set_zero(node->sub_tag);
      }
    }
    void push_up_(Node* node) { node->satisfied = node->ls->satisfied && node->rs->satisfied; }
    void add_(Node* node, i64 st, i64 ed, const Res& res) {
      if (node->ed <= st || ed <= node->st) return;
      if (st <= node->st && node->ed <= ed) { add_res_(node, res); return; }
      push_down_(node);
      add_(node->ls, st, ed, res);
      add_(node->rs, st, ed, res);
      push_up_(node);
    }
    void sub_(Node* node, i64 st, i64 ed, const Res& res) {
      if (node->ed <= st || ed <= node->st) return;
      if (st <= node->st && node->ed <= ed) { sub_res_(node, res); return; }
      push_down_(node);
      sub_(node->ls, st, ed, res);
      sub_(node->rs, st, ed, res);
      push_up_(node);
    }
    void destroy_(Node* node) {
      if (node == nullptr) return;
      destroy_(node->ls);
      destroy_(node->rs);
      delete node;
    }
    const A* A_;
    Res m_target_res_;
    Node* m_root_;

   public:
    PreemptSegTree(const A* alg, i64 st, i64 ed, const Res& target_res) : A_(alg), m_target_res_(target_res) {
      m_root_ = make(st, ed, false, nullptr);
    }
    ~PreemptSegTree() { destroy_(m_root_); }
    PreemptSegTree(const PreemptSegTree&) = delete;
    PreemptSegTree& operator=(const PreemptSegTree&) = delete;
    PreemptSegTree(PreemptSegTree&& o) noexcept : A_(o.A_), m_target_res_(o.m_target_res_), m_root_(o.m_root_) { o.m_root_ = nullptr; }
    void Add(i64 st, i64 ed, const Res& res) { add_(m_root_, st, ed, res); }
    void Sub(i64 st, i64 ed, const Res& res) { sub_(m_root_, st, ed, res); }
    bool IsSatisfied() const { return m_root_->satisfied; }
    u64 NodeCount() const { return count_(m_root_); }
   private:
    static u64 count_(const Node* n) { return n ? 1 + count_(n->ls) + count_(n->rs) : 0; }
  };

  // -- LocalScheduler::TryPreempt_, JobScheduler.cpp:6378-6505 ----------------------------------
  using JobRef = std::pair<bool, u32>;   // (is_pending, index)
  bool TryPreempt_(i64 now, PdJob* job, const std::vector<NodeState*>& nodes_to_sched) {
    if (job->qos >= pc_->qos_preempt.size() || pc_->qos_preempt[job->qos].empty()) return false;  // :6384-6385
    const auto& plist = pc_->qos_preempt[job->qos];
    std::set<JobRef> preemptable_set;  // :6387-6396
    for (const auto* node : nodes_to_sched)
      for (u32 qos : plist) {
        auto it = node->qos_job_map.find(qos);
        if (it != node->qos_job_map.end()) preemptable_set.insert(it->second.begin(), it->second.end());
      }
    if (preemptable_set.empty()) return false;
    std::vector<JobRef> ordered(preemptable_set.begin(), preemptable_set.end());  // :6399-6432
    auto is_preempting = [&](const JobRef& v) { return !v.first && pc_->preempting_set.count((*rn_)[v.second].job_id) != 0; };
    // The reference sorts a hash set's iteration order with an unstable sort: the order of candidates that compare
    // equal is unspecified there.  Canonical here: stable, on top of (pending first, ascending index).
    std::stable_sort(ordered.begin(), ordered.end(), [&](const JobRef& a, const JobRef& b) {
      bool ca = is_preempting(a), cb = is_preempting(b);
      if (ca != cb) return ca;
      bool pa = a.first, pb = b.first;
      if (pa != pb) return pa;
      if (pa) {
        const PdJob& x = (*pd_)[a.second]; const PdJob& y = (*pd_)[b.second];
        if (x.qos_priority != y.qos_priority) return x.qos_priority < y.qos_priority;
        return x.priority < y.priority;
      } else {
        const RnJob& x = (*rn_)[a.second]; const RnJob& y = (*rn_)[b.second];
        if (x.qos_priority != y.qos_priority) return x.qos_priority < y.qos_priority;
        return x.start_time > y.start_time;
      }
    });
    // absl::InfiniteFuture() (the last key of every time map) and anything beyond ~36 years stay "after everything"
    auto ticks = [&](i64 t) -> i64 {
      const i64 d = t - now;
      if (t == kInfiniteFuture || d > (INT64_MAX / 8) / kTicksPerSecond) return INT64_MAX / 4;
      if (d < -((INT64_MAX / 8) / kTicksPerSecond)) return -(INT64_MAX / 4);
      return d * kTicksPerSecond;
    };
    const i64 seg_start = 0, seg_end = ticks(now + job->time_limit);  // :6434-6435
    std::vector<PreemptSegTree> seg_trees;
    seg_trees.reserve(nodes_to_sched.size());
    std::map<u32, size_t> node_index;
    for (size_t i = 0; i < nodes_to_sched.size(); ++i) {  // :6441-6461
      auto* node = nodes_to_sched[i];
      node_index[node->idx] = i;
      auto alloc_it = job->allocated_res.find(node->idx);
      assert(alloc_it != job->allocated_res.end());
      seg_trees.emplace_back(&A_, seg_start, seg_end, A_.from_mask(alloc_it->second));
      auto& tree = seg_trees.back();
      const auto& tmap = node->time_avail_res_map;
      for (auto it = tmap.begin(); it != tmap.end();) {
        i64 st = ticks(it->first);
        auto nxt = std::next(it);
        i64 ed = (nxt == tmap.end() ? seg_end : ticks(nxt->first));
        tree.Add(st, ed, it->second);
        if (ed >= seg_end) break;
        it = nxt;
      }
    }
    auto apply_to_trees = [&](const JobRef& pre, bool add) {  // :6463-6477
      const i64 st = ticks(pre.first ? (*pd_)[pre.second].start_time : (*rn_)[pre.second].start_time);
      const i64 ed = ticks(pre.first ? (*pd_)[pre.second].end_time : (*rn_)[pre.second].end_time);
      auto one = [&](u32 cid, const MaskRes& res) {
        auto ni = node_index.find(cid);
        if (ni == node_index.end()) return;
        if (add) seg_trees[ni->second].Add(st, ed, A_.from_mask(res));
        else seg_trees[ni->second].Sub(st, ed, A_.from_mask(res));
      };
      if (pre.first) for (const auto& [cid, res] : (*pd_)[pre.second].allocated_res) one(cid, res);
      else for (const auto& al : rn_alloc_map(pre.second)) one(al.first, al.second);
    };
    auto all_satisfied = [&]() {
      for (const auto& t : seg_trees)
        if (!t.IsSatisfied()) return false;
      return true;
    };
    if (getenv("ORA_DEBUG_PREEMPT")) {
      fprintf(stderr, "TryPreempt job %u qos %u: %zu candidates:", job->job_id, job->qos, ordered.size());
      for (auto& o : ordered) fprintf(stderr, " (%s %u)", o.first ? "pd" : "rn", o.second);
      fprintf(stderr, "  sat0:");
      for (auto& t : seg_trees) fprintf(stderr, " %d", (int)t.IsSatisfied());
      fprintf(stderr, "\n");
    }
    int preempt_idx = -1;
    for (size_t i = 0; !all_satisfied() && i < ordered.size(); ++i) {  // :6484-6488
      apply_to_trees(ordered[i], /*add=*/true);
      preempt_idx = static_cast<int>(i);
    }
    if (!all_satisfied()) return false;
    // (:6490: ordered[preempt_idx] with preempt_idx == -1 when the trees are satisfied before anything was added is
    // undefined behaviour in the reference; it cannot happen: the job would have started now in GetNodesAndTrySchedule_
    // ... unless the allocation against res_total fits the window minimum where get_max_tasks' greedy split did not.
    // The oracle then preempts nobody.)
    if (preempt_idx >= 0) job->preempted_jobs.push_back(ordered[preempt_idx]);
    for (int i = preempt_idx - 1; i >= 0; --i) {  // :6491-6497
      apply_to_trees(ordered[i], /*add=*/false);
      if (!all_satisfied()) {
        apply_to_trees(ordered[i], /*add=*/true);
        job->preempted_jobs.push_back(ordered[i]);
      }
    }
    seg_tree_nodes_ = 0;
    for (const auto& t : seg_trees) seg_tree_nodes_ += t.NodeCount();
    job->craned_ids.clear();  // :6499-6503
    for (const auto* node : nodes_to_sched) job->craned_ids.emplace_back(node->idx);
    job->start_time = now;
    return true;
  }

  // -- LocalScheduler::UpdateNodeSelectorWithPreemptedJob, JobScheduler.h:645-670 -----------------
  void UpdateNodeSelectorWithPreemptedJob(LocalScheduler& ls, i64 now, const JobRef& pre) {
    if (!pre.first) {
      RnJob& rn = (*rn_)[pre.second];
      std::map<u32, MaskRes> res = rn_alloc_map(pre.second);
      ReleaseResourceIfPresent(ls, now, rn.end_time, res);
      for (const auto& [cid, _] : res) {
        auto it = ls.sel.m_node_info_map_.find(cid);
        if (it == ls.sel.m_node_info_map_.end()) continue;
        it->second.node_state->qos_job_map[rn.qos].erase(pre);
      }
    } else {
      PdJob& pd = (*pd_)[pre.second];
      ReleaseResourceIfPresent(ls, pd.start_time, pd.end_time, pd.allocated_res);
      if (pd.reason == CNS_REASON_NONE) {  // is_scheduled()
        for (u32 cid : pd.craned_ids) {
          auto it = ls.sel.m_node_info_map_.find(cid);
          if (it == ls.sel.m_node_info_map_.end()) continue;
          it->second.node_state->qos_job_map[pd.qos].erase(pre);
        }
      }
    }
  }

  // -- LocalScheduler::GetNodesAndTrySchedule_, JobScheduler.cpp:6147-6369 --------------
  struct node_info {
    int ntasks_on_node;
    Res res;
    NodeState* node_state;
    bool operator<(const node_info& other) const { return ntasks_on_node > other.ntasks_on_node; }
  };

  bool GetNodesAndTrySchedule_(LocalScheduler& ls, i64 now, PdJob* job,
                               std::vector<NodeState*>* nodes_to_sched) {
    i64 earliest_end_time = now + job->time_limit;
    const ReqView min_res_view = ComposeView(job->req_node, job->req_task, job->tpn_min);

    std::priority_queue<node_info> topk_nodes_total;
    int topk_ntasks_sum_total = 0;
    std::priority_queue<node_info> topk_nodes_avail;
    int topk_ntasks_sum_avail = 0;

    auto get_max_tasks = [&](const Res& res_on_node) {  // :6171-6186
      Res feasible_res;
      if (!A_.feasible(min_res_view, res_on_node, &feasible_res)) return 0;
      Res res_avail = res_on_node;
      A_.sub(res_avail, feasible_res);
      int ntasks_on_node = (int)job->tpn_min;
      while (ntasks_on_node < static_cast<int>(job->tpn_max) &&
             A_.feasible(job->req_task, res_avail, &feasible_res)) {
        ++ntasks_on_node;
        A_.sub(res_avail, feasible_res);
      }
      return ntasks_on_node;
    };

    for (const auto& [cost, node_idx] : ls.sel.m_cost_node_info_set_) {  // :6188
      NodeState* node_state = ls.sel.m_node_info_map_.at(node_idx).node_state;
      auto& time_avail_res_map = node_state->time_avail_res_map;
      if (time_avail_res_map.size() >= kAlgoMaxJobNumPerNode) continue;  // :6194
      if (!job->included_nodes.empty() && !job->included_nodes.count(node_idx)) continue;
      if (!job->excluded_nodes.empty() && job->excluded_nodes.count(node_idx)) continue;

      int ntasks_on_node_total = get_max_tasks(node_state->res_total);  // :6222
      if (ntasks_on_node_total == 0) continue;

      if (topk_nodes_total.size() < job->node_num ||
          (u32)topk_ntasks_sum_total < job->ntasks) {  // :6233-6242
        topk_ntasks_sum_total += ntasks_on_node_total;
        topk_nodes_total.push(node_info{ntasks_on_node_total, node_state->res_total, node_state});
        if (topk_nodes_total.size() > job->node_num) {
          topk_ntasks_sum_total -= topk_nodes_total.top().ntasks_on_node;
          topk_nodes_total.pop();
        }
      }

      if (job->exclusive) {  // :6249-6271
        bool satisfied = true;
        for (const auto& [time, res] : time_avail_res_map) {
          if (time >= earliest_end_time) break;
          if (!A_.le(node_state->res_total, res)) { satisfied = false; break; }
        }
        if (!satisfied) continue;
        topk_ntasks_sum_avail += ntasks_on_node_total;
        topk_nodes_avail.push(node_info{ntasks_on_node_total, node_state->res_total, node_state});
        if (topk_nodes_avail.size() > job->node_num) {
          topk_ntasks_sum_avail -= topk_nodes_avail.top().ntasks_on_node;
          topk_nodes_avail.pop();
        }
        if (topk_nodes_avail.size() == job->node_num &&
            (u32)topk_ntasks_sum_avail >= job->ntasks)
          break;
      } else {  // :6272-6299
        Res feasible_res;
        if (!A_.feasible(min_res_view, node_state->res_avail, &feasible_res)) continue;
        Res min_res_on_node = node_state->res_avail;
        for (const auto& [time, res] : time_avail_res_map) {
          if (time >= earliest_end_time) break;
          A_.ckmin(min_res_on_node, res);
        }
        int ntasks_on_node_avail = get_max_tasks(min_res_on_node);
        if (dbg_false_cand_ && !ntasks_on_node_avail) {
          // (diagnostics for the engine's front filter: the entry at `now` alone would have admitted this node)
          const Res& front = time_avail_res_map.begin()->second;
          Res fr;
          if (A_.feasible(min_res_view, front, &fr)) {
            ++dbg_false_total_;
            const MaskRes m = A_.to_mask(min_res_on_node), f = A_.to_mask(front);
            const ReqView& v = min_res_view;
            if (v.cpu > m.cpu) ++dbg_false_cpu_;
            else if (v.mem > m.mem) ++dbg_false_mem_;
            else if (__builtin_popcountll(m.gres) < __builtin_popcountll(f.gres)) ++dbg_false_gres_;
            else ++dbg_false_cores_;
          }
        }
        if (ntasks_on_node_avail) {
          topk_ntasks_sum_avail += ntasks_on_node_avail;
          topk_nodes_avail.push(node_info{ntasks_on_node_avail, min_res_on_node, node_state});
          if (topk_nodes_avail.size() > job->node_num) {
            topk_ntasks_sum_avail -= topk_nodes_avail.top().ntasks_on_node;
            topk_nodes_avail.pop();
          }
          if (topk_nodes_avail.size() == job->node_num &&
              (u32)topk_ntasks_sum_avail >= job->ntasks)
            break;
        }
      }
    }

    auto distribute = [&](std::priority_queue<node_info>& q, bool record_nodes) {
      int rest_ntasks = (int)job->ntasks - (int)job->node_num;  // :6304 / :6345
      while (!q.empty()) {
        const auto& info = q.top();
        const auto& res = info.res;
        int ntasks_on_node = std::min(rest_ntasks, info.ntasks_on_node - 1) + 1;
        u32 nid = info.node_state->idx;
        if (job->exclusive) {
          add_alloc(job, nid, A_.to_mask(res));
        } else {
          Res feasible_res;
          bool ok = A_.feasible(ComposeView(job->req_node, job->req_task, (u32)ntasks_on_node),
                                res, &feasible_res);
          assert(ok);
          (void)ok;
          add_alloc(job, nid, A_.to_mask(feasible_res));
        }
        job->craned_id_to_task_num[nid] = (u32)ntasks_on_node;
        if (record_nodes) nodes_to_sched->push_back(info.node_state);
        rest_ntasks -= ntasks_on_node - 1;
        q.pop();
      }
    };

    if (topk_nodes_avail.size() == job->node_num &&
        (u32)topk_ntasks_sum_avail >= job->ntasks) {  // :6302-6333
      distribute(topk_nodes_avail, false);
      job->start_time = now;
      job->craned_ids.clear();
      for (const auto& [nid, _] : job->craned_id_to_task_num) job->craned_ids.push_back(nid);
      return true;
    }
    if (topk_nodes_total.size() < job->node_num ||
        (u32)topk_ntasks_sum_total < job->ntasks)  // :6335-6343
      return false;
    distribute(topk_nodes_total, true);  // :6345-6367
    return false;
  }

  // -- EarliestStartSubsetSelector, JobScheduler.h:678-865 -----------------------------
  class TimeAvailResMapIter;
  class ResMapIterList {
   public:
    struct Node {
      TimeAvailResMapIter* res_map_it;
      i64 time;
      bool first_k;
    };
    using ListContainer = std::list<Node>;
    using iterator = typename ListContainer::iterator;
    explicit ResMapIterList(size_t k_value) : m_k_value_(k_value), m_kth_it_(m_tracker_list_.end()) {}
    void emplace_back(TimeAvailResMapIter* it, i64 time) {
      if (it->m_tracker_list_it_ != m_tracker_list_.end()) return;
      m_tracker_list_.push_back(Node{it, time, m_tracker_list_.size() < m_k_value_});
      if (m_tracker_list_.size() == m_k_value_) {
        assert(m_kth_it_ == m_tracker_list_.end());
        m_kth_it_ = std::prev(m_tracker_list_.end());
      }
      it->m_tracker_list_it_ = std::prev(m_tracker_list_.end());
    }
    void erase(TimeAvailResMapIter* it) {
      if (it->m_tracker_list_it_ == m_tracker_list_.end()) return;
      const Node& node = *it->m_tracker_list_it_;
      if (m_kth_it_ != m_tracker_list_.end() && node.first_k) {
        m_kth_it_ = std::next(m_kth_it_);
        if (m_kth_it_ != m_tracker_list_.end()) m_kth_it_->first_k = true;
      }
      m_tracker_list_.erase(it->m_tracker_list_it_);
      it->m_tracker_list_it_ = m_tracker_list_.end();
    }
    i64 KthTime() const {
      if (m_kth_it_ == m_tracker_list_.end()) return kInfiniteFuture;
      return m_kth_it_->time;
    }
    iterator Begin() { return m_tracker_list_.begin(); }
    iterator KthIterator() const { return m_kth_it_; }
    iterator End() { return m_tracker_list_.end(); }
    ListContainer m_tracker_list_;
   private:
    const size_t m_k_value_;
    iterator m_kth_it_;
  };

  class TimeAvailResMapIter {
   public:
    TimeAvailResMapIter(const A* alg, u32 craned_id, typename TimeAvailResMap::const_iterator it,
                        typename TimeAvailResMap::const_iterator end, ResMapIterList* tracker_list,
                        const Res* job_res)
        : m_tracker_list_it_(tracker_list->m_tracker_list_.end()),
          alg_(alg), job_res(job_res), m_it_(it), m_end_(end), m_craned_id_(craned_id) {
      m_satisfied_flag_ = Satisfied();
    }
    bool IsCurrentPosSatisfied() const { return m_satisfied_flag_; }
    bool ReachEnd() const { return m_it_ == m_end_; }
    void Advance() {
      m_satisfied_flag_ = !m_satisfied_flag_;
      if (m_satisfied_flag_) { while (++m_it_ != m_end_ && !Satisfied()); }
      else { while (++m_it_ != m_end_ && Satisfied()); }
    }
    i64 Time() const { return m_it_->first; }
    u32 GetCranedId() const { return m_craned_id_; }
    typename ResMapIterList::iterator m_tracker_list_it_;
   private:
    bool Satisfied() const { return alg_->le(*job_res, m_it_->second); }
    const A* alg_;
    const Res* job_res;
    typename TimeAvailResMap::const_iterator m_it_;
    const typename TimeAvailResMap::const_iterator m_end_;
    const u32 m_craned_id_;
    bool m_satisfied_flag_;
  };

  bool Backfill_(i64 now, PdJob* job, const std::vector<NodeState*>& nodes) {  // cpp:6371-6376
    ResMapIterList m_satisfied_iters_(job->node_num);
    auto cmp = [](const TimeAvailResMapIter* lhs, const TimeAvailResMapIter* rhs) {
      return lhs->Time() > rhs->Time();
    };
    std::priority_queue<TimeAvailResMapIter*, std::vector<TimeAvailResMapIter*>,
                        std::function<bool(const TimeAvailResMapIter*, const TimeAvailResMapIter*)>>
        m_time_priority_queue_(cmp);
    std::vector<TimeAvailResMapIter> m_res_map_iters_;
    std::vector<Res> job_res_store;
    m_res_map_iters_.reserve(nodes.size());
    job_res_store.reserve(nodes.size());
    for (const NodeState* node : nodes) {
      job_res_store.push_back(A_.from_mask(job->allocated_res.at(node->idx)));
      m_res_map_iters_.emplace_back(&A_, node->idx, node->time_avail_res_map.begin(),
                                    node->time_avail_res_map.end(), &m_satisfied_iters_,
                                    &job_res_store.back());
      m_time_priority_queue_.emplace(&m_res_map_iters_.back());
    }
    // CalcEarliestStartTime, JobScheduler.h:812-855
    while (!m_time_priority_queue_.empty()) {
      i64 current_time = m_time_priority_queue_.top()->Time();
      // absl::Duration subtraction saturates: InfiniteFuture - now is +inf > window.
      if (current_time == kInfiniteFuture || current_time - now > kAlgoMaxTimeWindow) return false;
      while (true) {
        if (m_time_priority_queue_.empty()) break;
        TimeAvailResMapIter* it = m_time_priority_queue_.top();
        if (it->Time() != current_time) break;
        m_time_priority_queue_.pop();
        if (it->IsCurrentPosSatisfied()) m_satisfied_iters_.emplace_back(it, current_time);
        else m_satisfied_iters_.erase(it);
        it->Advance();
        if (!it->ReachEnd()) m_time_priority_queue_.emplace(it);
      }
      i64 kth_time = m_satisfied_iters_.KthTime();
      if (kth_time == kInfiniteFuture) continue;
      // kth_time + time_limit <= top: evaluated in absl saturating arithmetic; an
      // InfiniteFuture top always satisfies it.
      if (m_time_priority_queue_.empty() ||
          m_time_priority_queue_.top()->Time() == kInfiniteFuture ||
          kth_time + job->time_limit <= m_time_priority_queue_.top()->Time()) {
        job->start_time = kth_time;
        job->craned_ids.clear();
        auto it = m_satisfied_iters_.Begin();
        while (true) {
          job->craned_ids.emplace_back(it->res_map_it->GetCranedId());
          if (it++ == m_satisfied_iters_.KthIterator()) break;
        }
        assert(job->craned_ids.size() == job->node_num);
        return true;
      }
    }
    return false;
  }

  // -- LocalScheduler::CalculateRunningNodesAndStartTime_, cpp:6127-6145 ------------------------
  bool CalculateRunningNodesAndStartTime_(LocalScheduler& ls, i64 now, PdJob* job) {
    std::vector<NodeState*> nodes_to_sched;
    if (GetNodesAndTrySchedule_(ls, now, job, &nodes_to_sched)) return true;
    if (nodes_to_sched.size() < job->node_num) return false;
    if (pc_ && pc_->enabled && TryPreempt_(now, job, nodes_to_sched)) return true;  // :6140-6143
    return Backfill_(now, job, nodes_to_sched);
  }

  // -- SchedulerAlgo::NodeSelect, cpp:6507-6836 ------------------------------------------
  // nodes: res_total per dense node index + schedulable flag; partitions: node lists.
  void NodeSelect(i64 now, const std::vector<MaskRes>& node_total,
                  const std::vector<uint8_t>& schedulable,
                  const std::vector<std::vector<u32>>& part_nodes, std::vector<RnJob>& running_jobs,
                  std::vector<PdJob>& pending_jobs, u64 scheduled_batch_size,
                  const std::vector<Resv>& resvs = {}, PreemptCfg* preempt = nullptr) {
    for (auto& rn : running_jobs) rn.end_time = std::max(rn.end_time, now + 1);  // :6513-6514
    pc_ = preempt; pd_ = &pending_jobs; rn_ = &running_jobs;
    if (pc_) {  // :6545-6559: running jobs that are being preempted end "now"; ids that left the running set are dropped
      std::map<u32, RnJob*> rn_job_id_map;
      for (auto& rn : running_jobs) rn_job_id_map.emplace(rn.job_id, &rn);
      for (auto it = pc_->preempting_set.begin(); it != pc_->preempting_set.end();) {
        auto rn_it = rn_job_id_map.find(*it);
        if (rn_it == rn_job_id_map.end()) it = pc_->preempting_set.erase(it);
        else { rn_it->second->end_time = now + 1; ++it; }
      }
      pc_->cancelled.clear();
    }
    // :6524-6530 reservations that have pending jobs (before ordering / batch limit)
    std::vector<char> resv_has_pd(resvs.size(), 0);
    for (const auto& job : pending_jobs)
      if (job.reservation != CNS_RESV_NONE && job.reservation < resvs.size()) resv_has_pd[job.reservation] = 1;

    node_state_.clear();
    node_state_.resize(node_total.size());
    for (u32 n = 0; n < node_total.size(); ++n) {  // :6584-6607
      if (!schedulable[n]) continue;
      auto ns = std::make_unique<NodeState>();
      ns->idx = n;
      ns->res_total = A_.from_mask(node_total[n]);
      ns->res_avail = ns->res_total;
      node_state_[n] = std::move(ns);
    }
    // :6619-6679 reservations (canonical order: ascending index; the reference iterates a hash map)
    first_resv_.assign(node_total.size(), kInfiniteFuture);
    resv_node_state_.clear();
    resv_node_state_.resize(resvs.size());
    resv_end_.assign(resvs.size(), 0);
    resv_live_.assign(resvs.size(), 0);
    for (u32 v = 0; v < resvs.size(); ++v) {
      const Resv& rv = resvs[v];
      if (now >= rv.end_time) continue;  // expired
      for (const auto& al : rv.allocs)
        if (al.node < first_resv_.size()) first_resv_[al.node] = std::min(first_resv_[al.node], rv.start_time);
      if (now >= rv.start_time) {
        for (const auto& al : rv.allocs)
          if (al.node < node_state_.size() && node_state_[al.node])
            node_state_[al.node]->allocated_res.push_back({rv.end_time, A_.from_mask(al.res)});
        if (!resv_has_pd[v]) continue;  // no pending jobs, skip
        resv_live_[v] = 1;
        resv_end_[v] = rv.end_time;
        for (const auto& al : rv.allocs) {
          auto ns = std::make_unique<NodeState>();
          ns->idx = al.node;
          ns->res_total = A_.from_mask(al.res);
          ns->res_avail = ns->res_total;
          resv_node_state_[v].emplace(al.node, std::move(ns));
        }
      } else {
        for (const auto& al : rv.allocs)
          if (al.node < node_state_.size() && node_state_[al.node])
            node_state_[al.node]->reserved_res.push_back({rv.start_time, rv.end_time, A_.from_mask(al.res)});
      }
    }
    for (u32 ri = 0; ri < running_jobs.size(); ++ri) {  // :6681-6709
      const auto& job = running_jobs[ri];
      if (job.reservation == CNS_RESV_NONE) {
        for (const auto& al : job.allocs)
          if (al.node < node_state_.size() && node_state_[al.node]) {
            node_state_[al.node]->allocated_res.push_back({job.end_time, A_.from_mask(al.res)});
            node_state_[al.node]->qos_job_map[job.qos].emplace(false, ri);  // :6688
          }
      } else {
        if (job.reservation >= resvs.size() || !resv_live_[job.reservation]) continue;  // "not found" (:6693-6700)
        auto& m = resv_node_state_[job.reservation];
        for (const auto& al : job.allocs) {
          auto it = m.find(al.node);
          if (it != m.end()) {
            it->second->allocated_res.push_back({job.end_time, A_.from_mask(al.res)});
            it->second->qos_job_map[job.qos].emplace(false, ri);  // :6705
          }
        }
      }
    }

    for (auto& ns : node_state_)  // :6712-6714
      if (ns) InitTimeAvailResMap(*ns, now);
    for (u32 v = 0; v < resvs.size(); ++v)  // :6715-6719
      for (auto& [nid, ns] : resv_node_state_[v]) InitTimeAvailResMap(*ns, now, resv_end_[v]);

    part_scheduler_.clear();
    part_scheduler_.resize(part_nodes.size());
    for (size_t p = 0; p < part_nodes.size(); ++p)  // :6723-6728 (JobScheduler.h:603-612)
      for (u32 n : part_nodes[p])
        if (node_state_[n]) AddNode(part_scheduler_[p], now, node_state_[n].get());
    resv_scheduler_.clear();
    resv_scheduler_.resize(resvs.size());
    for (u32 v = 0; v < resvs.size(); ++v)  // :6729-6732
      for (auto& [nid, ns] : resv_node_state_[v]) AddNode(resv_scheduler_[v], now, ns.get());

    // BasicPriority::GetOrderedJobPtrVec, JobScheduler.h:185-200
    size_t len = pending_jobs.size();
    if (scheduled_batch_size) len = std::min<size_t>(len, scheduled_batch_size);
    for (size_t i = len; i < pending_jobs.size(); ++i) pending_jobs[i].reason = CNS_REASON_PRIORITY;

    jobs_ordered_ = len;
    for (size_t i = 0; i < len; ++i) {  // :6743-6835
      PdJob* job = &pending_jobs[i];
      if (job->skip) { job->reason = CNS_REASON_SKIPPED; continue; }  // :6744
      LocalScheduler* sched_ptr;
      if (job->reservation == CNS_RESV_NONE) {
        if (job->partition >= part_scheduler_.size()) {  // :6748-6752
          job->reason = CNS_REASON_PARTITION_NOT_FOUND;
          continue;
        }
        sched_ptr = &part_scheduler_[job->partition];
      } else {
        if (job->reservation >= resv_scheduler_.size() || !resv_live_[job->reservation]) {  // :6754-6759
          job->reason = CNS_REASON_RESERVATION_NOT_FOUND;
          continue;
        }
        sched_ptr = &resv_scheduler_[job->reservation];
      }
      LocalScheduler& scheduler = *sched_ptr;
      bool ok = CalculateRunningNodesAndStartTime_(scheduler, now, job);
      if (!ok) {
        job->reason = CNS_REASON_RESOURCE;  // :6768
        job->start_time = 0;
        job->allocated_res.clear();
        job->craned_id_to_task_num.clear();
        job->craned_ids.clear();
      } else {
        job->end_time = job->start_time + job->time_limit;  // :6772
        for (const auto& pre : job->preempted_jobs) {  // :6779-6792
          UpdateNodeSelectorWithPreemptedJob(scheduler, now, pre);
          if (pre.first) { pending_jobs[pre.second].reason = CNS_REASON_PREEMPTED; continue; }
          const u32 rid = running_jobs[pre.second].job_id;
          if (!pc_->preempting_set.insert(rid).second) continue;
          pc_->cancelled.push_back(rid);   // EnqueuePreemptCancel (:6793)
        }
        AllocateResource(scheduler, job->start_time, job->end_time, job->allocated_res);  // :6795 (h:630-643)
        if (job->reason == CNS_REASON_NONE)   // is_scheduled(): the reason of a later start is only set below
          for (u32 nid : job->craned_ids) {
            auto nit = scheduler.sel.m_node_info_map_.find(nid);
            nit->second.node_state->qos_job_map[job->qos].emplace(true, (u32)i);
          }
        if (job->start_time != now) {  // :6797-6833
          if (job->reservation == CNS_RESV_NONE) {
            for (u32 nid : job->craned_ids)
              if (first_resv_[nid] != kInfiniteFuture && first_resv_[nid] < now + job->time_limit) {  // :6799-6806
                job->reason = CNS_REASON_RESOURCE_RESERVED;
                break;
              }
            if (job->reason != CNS_REASON_NONE) continue;
            for (u32 nid : job->craned_ids) {
              const Res& res_avail = node_state_[nid]->res_avail;
              if (!A_.le(A_.from_mask(job->allocated_res.at(nid)), res_avail)) {
                job->reason = CNS_REASON_RESOURCE;
                break;
              }
            }
          } else {
            for (u32 nid : job->craned_ids) {
              const Res& res_avail = resv_node_state_[job->reservation].at(nid)->res_avail;
              if (!A_.le(A_.from_mask(job->allocated_res.at(nid)), res_avail)) {
                job->reason = CNS_REASON_RESOURCE;
                break;
              }
            }
          }
          if (job->reason == CNS_REASON_NONE) job->reason = CNS_REASON_PRIORITY;
        }
      }
    }
  }

  double CostOf(u32 part, u32 node) const { return part_scheduler_[part].sel.m_node_info_map_.at(node).cost; }
  bool HasNode(u32 node) const { return node < node_state_.size() && node_state_[node] != nullptr; }
  const TimeAvailResMap& Timeline(u32 node) const { return node_state_[node]->time_avail_res_map; }
  bool HasResvNode(u32 v, u32 node) const { return v < resv_node_state_.size() && resv_node_state_[v].count(node) != 0; }
  const TimeAvailResMap& ResvTimeline(u32 v, u32 node) const { return resv_node_state_[v].at(node)->time_avail_res_map; }
  double ResvCostOf(u32 v, u32 node) const { return resv_scheduler_[v].sel.m_node_info_map_.at(node).cost; }
  const A& alg() const { return A_; }
  u64 jobs_ordered() const { return jobs_ordered_; }
  u64 last_seg_tree_nodes() const { return seg_tree_nodes_; }

 private:
  std::map<u32, MaskRes> rn_alloc_map(u32 ri) const {  // RnJobInScheduler::allocated_res.EachNodeResMap()
    std::map<u32, MaskRes> m;
    for (const auto& al : (*rn_)[ri].allocs) {
      MaskRes& d = m[al.node];
      d.cpu += al.res.cpu; d.mem += al.res.mem; d.clo |= al.res.clo; d.chi |= al.res.chi; d.c2 |= al.res.c2; d.c3 |= al.res.c3; d.gres |= al.res.gres;
    }
    return m;
  }
  PreemptCfg* pc_ = nullptr;
  std::vector<PdJob>* pd_ = nullptr;
  std::vector<RnJob>* rn_ = nullptr;
  u64 seg_tree_nodes_ = 0;
  i64 cpu_of(const MaskRes& r) const { return r.cpu; }
  i64 cpu_of(const LitRes& r) const { return r.cpu; }
  static void add_alloc(PdJob* job, u32 nid, const MaskRes& r) {
    // ResourceV3::AddResourceInNode (PublicHeader.cpp:915-919): += into a fresh entry.
    MaskRes& d = job->allocated_res[nid];
    d.cpu += r.cpu; d.mem += r.mem; d.clo |= r.clo; d.chi |= r.chi; d.c2 |= r.c2; d.c3 |= r.c3; d.gres |= r.gres;
  }

  A A_;
  const u32 kAlgoMaxJobNumPerNode;
  const i64 kAlgoMaxTimeWindow;
  std::vector<std::unique_ptr<NodeState>> node_state_;
  std::vector<LocalScheduler> part_scheduler_;
  // reservations: per reservation its own NodeStates (res_total = the reserved share) and scheduler
  std::vector<std::map<u32, std::unique_ptr<NodeState>>> resv_node_state_;
  std::vector<LocalScheduler> resv_scheduler_;
  std::vector<i64> resv_end_, first_resv_;
  std::vector<char> resv_live_;
  u64 jobs_ordered_ = 0;
 public:
  bool dbg_false_cand_ = getenv("ORA_DEBUG_FALSE_CAND") != nullptr;
  u64 dbg_false_total_ = 0, dbg_false_cpu_ = 0, dbg_false_mem_ = 0, dbg_false_gres_ = 0, dbg_false_cores_ = 0;
  ~SchedOracle() {
    if (dbg_false_cand_)
      fprintf(stderr, "oracle: nodes the front entry admits but the window minimum rejects: %llu (cpu %llu, mem %llu, gres %llu, core ids %llu)\n",
              (unsigned long long)dbg_false_total_, (unsigned long long)dbg_false_cpu_, (unsigned long long)dbg_false_mem_,
              (unsigned long long)dbg_false_gres_, (unsigned long long)dbg_false_cores_);
  }
};

}  // namespace ora
