// ORACLE — TEST INFRASTRUCTURE ONLY (see res_algebra.hpp header). "Parity unpinned":
// the reference ships no test of NodeSelect; this file restates it line by line.
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
// Out of this slice, as in SURVEY.md §8: reservations, preemption, licenses.
// Compile with -ffp-contract=off: the fp64 cost must not be FMA-contracted.
#pragma once
#include <cstdint>
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
  // results
  i64 start_time = 0, end_time = 0;
  int reason = CNS_REASON_NONE;
  std::map<u32, MaskRes> allocated_res;          // ResourceV3 (node -> ResourceInNodeV3)
  std::map<u32, u32> craned_id_to_task_num;
  std::vector<u32> craned_ids;
};

struct RnAlloc { u32 node; MaskRes res; };
struct RnJob { i64 end_time; std::vector<RnAlloc> allocs; u32 reservation = CNS_RESV_NONE; };
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

  // -- NodeState::UpdateResourceInNode, JobScheduler.h:340-459 (allocate only) ---------
  void UpdateResourceInNode(NodeState& ns, i64 start_time, i64 end_time, const Res& res) {
    auto& m = ns.time_avail_res_map;
    bool ok;
    auto job_duration_begin_it = m.upper_bound(start_time);
    if (job_duration_begin_it == m.end()) {
      --job_duration_begin_it;
      typename TimeAvailResMap::iterator inserted_it;
      std::tie(inserted_it, ok) = m.emplace(end_time, job_duration_begin_it->second);
      assert(ok);
      if (job_duration_begin_it->first == start_time) {  // Case #1
        assert(A_.le(res, job_duration_begin_it->second));
        A_.sub(job_duration_begin_it->second, res);
      } else {  // Case #2
        std::tie(inserted_it, ok) = m.emplace(start_time, job_duration_begin_it->second);
        assert(ok);
        assert(A_.le(res, inserted_it->second));
        A_.sub(inserted_it->second, res);
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
      for (auto it = job_duration_begin_it; it != job_duration_end_it; it++) {
        assert(A_.le(res, it->second));
        A_.sub(it->second, res);
      }
      if (job_duration_end_it->first != end_time) {
        typename TimeAvailResMap::iterator inserted_it;
        std::tie(inserted_it, ok) = m.emplace(end_time, job_duration_end_it->second);
        assert(ok);
        assert(A_.le(res, job_duration_end_it->second));
        A_.sub(job_duration_end_it->second, res);
      }
    }
  }

  // -- MinCpuTimeRatioFirst::UpdateCost, JobScheduler.h:43-54 --------------------------
  static void UpdateCost(double& cost, i64 start_time, i64 end_time, i64 res_cpu_raw,
                         i64 total_cpu_raw) {
    // static_cast<double>(cpu_t) == double(raw) / 256.0 (fpm)
    double a = static_cast<double>(res_cpu_raw) / 256.0;
    double t = static_cast<double>(total_cpu_raw) / 256.0;
    double ratio = a / t;
    double delta = static_cast<double>(end_time - start_time) * ratio;
    cost += delta;
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

  // -- LocalScheduler::CalculateRunningNodesAndStartTime_, cpp:6127-6145 (no preemption) -
  bool CalculateRunningNodesAndStartTime_(LocalScheduler& ls, i64 now, PdJob* job) {
    std::vector<NodeState*> nodes_to_sched;
    if (GetNodesAndTrySchedule_(ls, now, job, &nodes_to_sched)) return true;
    if (nodes_to_sched.size() < job->node_num) return false;
    return Backfill_(now, job, nodes_to_sched);
  }

  // -- SchedulerAlgo::NodeSelect, cpp:6507-6836 ------------------------------------------
  // nodes: res_total per dense node index + schedulable flag; partitions: node lists.
  void NodeSelect(i64 now, const std::vector<MaskRes>& node_total,
                  const std::vector<uint8_t>& schedulable,
                  const std::vector<std::vector<u32>>& part_nodes, std::vector<RnJob>& running_jobs,
                  std::vector<PdJob>& pending_jobs, u64 scheduled_batch_size,
                  const std::vector<Resv>& resvs = {}) {
    for (auto& rn : running_jobs) rn.end_time = std::max(rn.end_time, now + 1);  // :6513-6514
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
    for (const auto& job : running_jobs) {  // :6681-6709
      if (job.reservation == CNS_RESV_NONE) {
        for (const auto& al : job.allocs)
          if (al.node < node_state_.size() && node_state_[al.node])
            node_state_[al.node]->allocated_res.push_back({job.end_time, A_.from_mask(al.res)});
      } else {
        if (job.reservation >= resvs.size() || !resv_live_[job.reservation]) continue;  // "not found" (:6693-6700)
        auto& m = resv_node_state_[job.reservation];
        for (const auto& al : job.allocs) {
          auto it = m.find(al.node);
          if (it != m.end()) it->second->allocated_res.push_back({job.end_time, A_.from_mask(al.res)});
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
        AllocateResource(scheduler, job->start_time, job->end_time, job->allocated_res);  // :6795
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

 private:
  i64 cpu_of(const MaskRes& r) const { return r.cpu; }
  i64 cpu_of(const LitRes& r) const { return r.cpu; }
  static void add_alloc(PdJob* job, u32 nid, const MaskRes& r) {
    // ResourceV3::AddResourceInNode (PublicHeader.cpp:915-919): += into a fresh entry.
    MaskRes& d = job->allocated_res[nid];
    d.cpu += r.cpu; d.mem += r.mem; d.clo |= r.clo; d.chi |= r.chi; d.gres |= r.gres;
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
};

}  // namespace ora
