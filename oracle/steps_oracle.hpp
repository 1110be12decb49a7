// CPU ORACLE (TEST INFRASTRUCTURE ONLY — never linked into or called by the product path).
//
// Restatement of JobInCtld::SchedulePendingSteps (src/CraneCtld/CtldPublicDefs.cpp:2038-2159), statement by statement,
// with the reference's std::priority_queue (so libstdc++'s heap decides ties) and the resource algebra of
// res_algebra.hpp.  One canonicalisation: the job's nodes are walked in the order given (the reference walks an
// unordered_map).  PINNED (round 4): the reference has no test of this function, but the function itself runs here:
// oracle/_ref compiles CtldPublicDefs.cpp:2038-2159 (sliced at build time by oracle/ref_build/extract.py) and
// tests/test_ref_pin_limits_steps.py holds this restatement to it (scheduled flags, nodes in pop order, every task's
// allocation, step_res_avail_ afterwards) on the hand-derived scenarios of tests/test_steps.py and 15 random cases.
#pragma once
#include <queue>
#include <set>
#include <vector>

#include "res_algebra.hpp"

namespace ora {

struct StepReq {
  ReqView node_view, task_view;   // req_node_res_view, req_task_res_view
  u32 node_num = 1, ntasks = 1, tmin = 1, tmax = 1;
  std::vector<u32> included, excluded;   // dense node indices
};
struct StepOut {
  bool scheduled = false;
  std::vector<u32> node;            // pop order
  std::vector<u32> node_ntasks;
  std::vector<MaskRes> node_alloc;  // step_alloc_res per node
  std::vector<u32> task_node;       // by task id
  std::vector<MaskRes> task_alloc;  // task_res_map
};

// nodes / avail: the job's allocation (step_res_avail_), updated in place.  Returns the outputs of the steps in order;
// steps after the first one that does not fit are left unscheduled (:2104-2106).
template <class Alg>
std::vector<StepOut> SchedulePendingSteps(const Alg& alg, const std::vector<u32>& nodes, std::vector<typename Alg::Res>& avail,
                                          const std::vector<StepReq>& steps) {
  using Res = typename Alg::Res;
  std::vector<StepOut> outs(steps.size());
  for (size_t si = 0; si < steps.size(); ++si) {
    const StepReq& step = steps[si];
    StepOut& out = outs[si];
    const u32 max_ntask_per_node = step.tmax, min_ntask_per_node = step.tmin;   // :2052-2053
    struct NodeInfo {                                                           // :2056-2062
      u32 ntasks_on_node;
      u32 pos;   // index into nodes / avail (const CranedId*)
      bool operator<(const NodeInfo& other) const { return ntasks_on_node > other.ntasks_on_node; }
    };
    std::priority_queue<NodeInfo> candidates;
    u32 sum_ntasks = 0;
    const std::set<u32> incl(step.included.begin(), step.included.end()), excl(step.excluded.begin(), step.excluded.end());
    for (u32 pos = 0; pos < nodes.size(); ++pos) {                              // :2066-2102
      if (excl.count(nodes[pos])) continue;
      if (!incl.empty() && !incl.count(nodes[pos])) continue;
      Res feasible_res;
      if (!alg.feasible(step.node_view, avail[pos], &feasible_res)) continue;
      Res res_avail = avail[pos];
      alg.sub(res_avail, feasible_res);
      u32 ntasks_on_node = 0;
      while (ntasks_on_node < max_ntask_per_node && alg.feasible(step.task_view, res_avail, &feasible_res)) {
        ++ntasks_on_node;
        alg.sub(res_avail, feasible_res);
      }
      if (ntasks_on_node < min_ntask_per_node) continue;
      candidates.push(NodeInfo{ntasks_on_node, pos});
      sum_ntasks += ntasks_on_node;
      if (candidates.size() > step.node_num) {
        sum_ntasks -= candidates.top().ntasks_on_node;
        candidates.pop();
      }
      if (candidates.size() == step.node_num && sum_ntasks >= step.ntasks) break;
    }
    if (candidates.size() < step.node_num || sum_ntasks < step.ntasks) break;   // :2104-2106
    u32 rest_ntasks = step.ntasks - step.node_num;                              // :2107
    while (!candidates.empty()) {                                               // :2109-2128
      const NodeInfo info = candidates.top();
      Res& res_avail = avail[info.pos];
      Res feasible_res, node_sum;
      alg.set_zero(node_sum);
      alg.feasible(step.node_view, res_avail, &feasible_res);
      alg.sub(res_avail, feasible_res);
      alg.add(node_sum, feasible_res);
      const u32 ntasks_on_node = std::min(rest_ntasks, info.ntasks_on_node - 1) + 1;
      for (u32 i = 0; i < ntasks_on_node; ++i) {
        alg.feasible(step.task_view, res_avail, &feasible_res);
        alg.sub(res_avail, feasible_res);
        out.task_node.push_back(nodes[info.pos]);
        out.task_alloc.push_back(alg.to_mask(feasible_res));
        alg.add(node_sum, feasible_res);
      }
      rest_ntasks -= ntasks_on_node - 1;
      out.node.push_back(nodes[info.pos]);
      out.node_ntasks.push_back(ntasks_on_node);
      out.node_alloc.push_back(alg.to_mask(node_sum));
      candidates.pop();
    }
    out.scheduled = true;
  }
  return outs;
}

}  // namespace ora
