// TEST INFRASTRUCTURE ONLY — CPU restatement of MultiFactorPriority, the checker of cns_priority_order.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
// PINNED (round 3): the reference ships no test or golden vector for this class, but its own
// MultiFactorPriority::{GetOrderedJobPtrVec, CalculateFactorBound_, CalculatePriority_} are compiled into oracle/_ref
// (oracle/ref_build/extract.py) and tests/test_ref_pin.py holds this restatement to them (fp64 priorities as bit
// patterns and the order, 40 random cases), beside the hand-derived KATs of tests/test_priority.py.
//
// Reference: src/CraneCtld/JobScheduler.cpp:7606-7819 (GetOrderedJobPtrVec :7606-7631,
// CalculateFactorBound_ :7633-7752, CalculatePriority_ :7754-7817), struct FactorBound JobScheduler.h:214-224,
// PriorityConfig CtldPublicDefs.h:162-174.  cpu_t = fpm::fixed<int64,__int128,8>: CpuCountDouble() is
// static_cast<double>(raw) / 256 (PublicHeader.cpp:509-511).
//
// Canonicalisation: std::ranges::sort is unstable -> stable sort, ties by ascending input
// index; RnJobInScheduler::node_num (uninitialised in the reference) is an input.
#pragma once
#include <algorithm>
#include <cstdint>
#include <limits>
#include <numeric>
#include <vector>

namespace ora {

struct PrioConfig {
  uint64_t max_age;
  uint32_t w_age, w_fair, w_size, w_part, w_qos;
  bool favor_small;
};
struct PrioPending {
  int64_t submit;
  uint32_t qos, part, node_num;
  int64_t cpu_raw;
  uint64_t mem;
  uint32_t account;
  double cached;
};
struct PrioRunning {
  int64_t start;
  uint32_t qos, part, node_num;
  int64_t cpu_raw;
  uint64_t mem;
  uint32_t account;
};

inline double cpu_double(int64_t raw) { return static_cast<double>(raw) / 256.0; }

// Returns the order; prio[j] = priority of pending job j.
inline std::vector<uint32_t> priority_order(int64_t now, const PrioConfig& cfg, uint32_t num_accounts,
                                            const std::vector<PrioPending>& pd, const std::vector<PrioRunning>& rn,
                                            std::vector<double>& prio) {
  // ---- CalculateFactorBound_ (:7633-7752) ----
  uint64_t age_max = 0, age_min = std::numeric_limits<uint64_t>::max();
  uint32_t qos_max = 0, qos_min = std::numeric_limits<uint32_t>::max();
  uint32_t part_max = 0, part_min = std::numeric_limits<uint32_t>::max();
  uint32_t nn_max = 0, nn_min = std::numeric_limits<uint32_t>::max();
  uint64_t mem_max = 0, mem_min = std::numeric_limits<uint64_t>::max();
  double cpus_max = 0, cpus_min = std::numeric_limits<double>::max();
  double sv_max = 0, sv_min = std::numeric_limits<uint32_t>::max();
  std::vector<double> acc_val(num_accounts, 0.0);
  std::vector<char> acc_present(num_accounts, 0);

  for (const auto& j : pd) {  // :7663-7690
    uint64_t age = static_cast<uint64_t>(now - j.submit);
    age = std::min(age, cfg.max_age);
    acc_val[j.account] = 0.0;
    acc_present[j.account] = 1;
    age_min = std::min(age, age_min); age_max = std::max(age, age_max);
    nn_min = std::min(j.node_num, nn_min); nn_max = std::max(j.node_num, nn_max);
    mem_min = std::min(j.mem, mem_min); mem_max = std::max(j.mem, mem_max);
    const double c = cpu_double(j.cpu_raw);
    cpus_min = std::min(c, cpus_min); cpus_max = std::max(c, cpus_max);
    qos_min = std::min(j.qos, qos_min); qos_max = std::max(j.qos, qos_max);
    part_min = std::min(j.part, part_min); part_max = std::max(j.part, part_max);
  }
  for (const auto& j : rn) {  // :7692-7713
    nn_min = std::min(j.node_num, nn_min); nn_max = std::max(j.node_num, nn_max);
    mem_min = std::min(j.mem, mem_min); mem_max = std::max(j.mem, mem_max);
    const double c = cpu_double(j.cpu_raw);
    cpus_min = std::min(c, cpus_min); cpus_max = std::max(c, cpus_max);
    qos_min = std::min(j.qos, qos_min); qos_max = std::max(j.qos, qos_max);
    part_min = std::min(j.part, part_min); part_max = std::max(j.part, part_max);
  }
  for (const auto& j : rn) {  // :7715-7746
    double service_val = 0;
    if (cpus_max > cpus_min)
      service_val += 1.0 * (cpu_double(j.cpu_raw) - cpus_min) / (cpus_max - cpus_min);
    else
      service_val += 1.0;
    if (nn_max > nn_min)
      service_val += 1.0 * (j.node_num - nn_min) / (nn_max - nn_min);
    else
      service_val += 1.0;
    if (mem_max > mem_min)
      service_val += 1.0 * static_cast<double>(j.mem - mem_min) / static_cast<double>(mem_max - mem_min);
    else
      service_val += 1.0;
    const uint64_t run_time = static_cast<uint64_t>(now - j.start);
    acc_val[j.account] += service_val * static_cast<double>(run_time);
    acc_present[j.account] = 1;
  }
  for (uint32_t a = 0; a < num_accounts; ++a) {  // :7748-7751
    if (!acc_present[a]) continue;
    sv_min = std::min(acc_val[a], sv_min);
    sv_max = std::max(acc_val[a], sv_max);
  }

  // ---- CalculatePriority_ (:7754-7817) for jobs without a cached priority (:7616) ----
  prio.assign(pd.size(), 0.0);
  for (size_t i = 0; i < pd.size(); ++i) {
    const auto& j = pd[i];
    if (j.cached != 0.0) { prio[i] = j.cached; continue; }
    uint64_t job_age = static_cast<uint64_t>(now - j.submit);
    job_age = std::min(job_age, cfg.max_age);
    double qos_factor{0}, age_factor{0}, partition_factor{0}, job_size_factor{0}, fair_share_factor{0};
    if (age_max > age_min)
      age_factor = 1.0 * static_cast<double>(job_age - age_min) / static_cast<double>(age_max - age_min);
    if (qos_max > qos_min) qos_factor = 1.0 * (j.qos - qos_min) / (qos_max - qos_min);
    if (part_max > part_min) partition_factor = 1.0 * (j.part - part_min) / (part_max - part_min);
    if (cpus_max > cpus_min) job_size_factor += 1.0 * (cpu_double(j.cpu_raw) - cpus_min) / (cpus_max - cpus_min);
    if (nn_max > nn_min) job_size_factor += 1.0 * (j.node_num - nn_min) / (nn_max - nn_min);
    if (mem_max > mem_min)
      job_size_factor += 1.0 * static_cast<double>(j.mem - mem_min) / static_cast<double>(mem_max - mem_min);
    if (cfg.favor_small)
      job_size_factor = 1.0 - job_size_factor / 3;
    else
      job_size_factor /= 3.0;
    if (sv_max > sv_min) fair_share_factor = 1.0 - (acc_val[j.account] - sv_min) / (sv_max - sv_min);
    prio[i] = cfg.w_age * age_factor + cfg.w_part * partition_factor + cfg.w_size * job_size_factor +
              cfg.w_fair * fair_share_factor + cfg.w_qos * qos_factor;
  }

  // ---- order (:7622-7624), canonical tie-break ----
  std::vector<uint32_t> order(pd.size());
  std::iota(order.begin(), order.end(), 0u);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return prio[a] > prio[b]; });
  return order;
}

}  // namespace ora
