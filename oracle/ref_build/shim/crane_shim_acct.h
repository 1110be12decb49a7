// ORACLE / TEST INFRASTRUCTURE ONLY — part 3 of the environment of the sliced REFERENCE sources
// (see crane_shim.h): what the run-limit admission (src/CraneCtld/Accounting/AccountMetaContainer.{h,cpp}) and
// the step scheduler (JobInCtld::SchedulePendingSteps, src/CraneCtld/CtldPublicDefs.cpp:2038-2159) read from
// the rest of CraneCtld.  Stand-ins written for this build with the member names the slices use; the reference
// definitions they replace are cited.  Included after crane_shim_ctld.h (Qos, AccountManager, JobInCtld) and the
// JobScheduler.h slice (PdJobInScheduler), before the AccountMetaContainer.h slice.
#pragma once
#include <mutex>
#include <shared_mutex>

#include "crane_shim_ctld.h"

// crane/PublicHeader.h:51,165
inline const cpu_t kUnlimitedCpu = cpu_t::from_raw_value(int64_t{1} << 53);
constexpr uint64_t kMaxJobMemoryBytes = 10737418240000;

// crane/PublicHeader.h: CraneErrCode is protobuf's crane::grpc::ErrCode; only named in declarations of the
// submit-time members of AccountMetaContainer, which the slices never define or call.
enum class CraneErrCode { SUCCESS = 0 };

// parallel-hashmap (dependencies/cmake/parallel-hashmap): the three members AccountMetaContainer's run path calls
// (AccountMetaContainer.cpp:895,901,908 contains; :935,965,984 if_contains; :1089,1109,1121 try_emplace_l).
// phmap's contract: if_contains / try_emplace_l run the functor on the element under the submap's lock and
// try_emplace_l constructs the mapped value from the trailing arguments when the key is absent.  Iteration order
// is never used by the run path, so an ordered map is an exact stand-in.
namespace phmap {
namespace priv {
template <class T> struct hash_default_hash {};
template <class T> struct hash_default_eq {};
}  // namespace priv
template <class K, class V, class H = void, class E = void, class A = void, size_t N = 4, class M = void>
class parallel_flat_hash_map {
  std::map<K, V> m_;

 public:
  using value_type = std::pair<const K, V>;
  bool contains(const K& k) const { return m_.count(k) != 0; }
  template <class F>
  bool if_contains(const K& k, F&& f) {
    auto it = m_.find(k);
    if (it == m_.end()) return false;
    std::forward<F>(f)(*it);
    return true;
  }
  template <class F, class... Args>
  bool try_emplace_l(const K& k, F&& f, Args&&... args) {
    auto it = m_.find(k);
    if (it != m_.end()) {
      std::forward<F>(f)(*it);
      return false;
    }
    m_.emplace(std::piecewise_construct, std::forward_as_tuple(k), std::forward_as_tuple(std::forward<Args>(args)...));
    return true;
  }
  // harness access (seeding the usage the commit loop starts from, reading it back)
  V& operator[](const K& k) { return m_[k]; }
  auto find(const K& k) const { return m_.find(k); }
  auto begin() const { return m_.begin(); }
  auto end() const { return m_.end(); }
};
}  // namespace phmap

namespace util {
// crane/String.h: only feeds allocated_craneds_regex (a display string), CtldPublicDefs.cpp:2136
template <class C>
inline std::string HostNameListToStr(const C&) { return {}; }
}  // namespace util

namespace Ctld {

// ---- step scheduler: src/CraneCtld/CtldPublicDefs.h ------------------------------------------------------
// StepInteractiveMeta (:281-361): the callback SchedulePendingSteps fires when a step got its allocation (:2143-2155)
struct StepInteractiveMeta {
  struct StepResAllocArgs {
    job_id_t job_id;
    step_id_t step_id;
    struct ResAllocInfo {
      std::string allocated_craned_regex;
      std::vector<CranedId> allocated_craned_ids;
      std::unordered_map<CranedId, std::set<task_id_t>> craned_task_map;
      uint32_t ntasks_total;
    };
    std::expected<ResAllocInfo, std::string> res_allocate_expt;
  };
  std::function<void(StepResAllocArgs const&)> cb_step_res_allocated;
};

// CommonStepInCtld / StepInCtld (:533-790): the members SchedulePendingSteps reads and writes
struct CommonStepInCtld {
  job_id_t job_id{0};
  step_id_t step_id{0};
  uint32_t ntasks_per_node_min{1};
  uint32_t ntasks_per_node_max{1};
  uint32_t node_num{1};
  uint32_t ntasks{1};
  std::unordered_set<std::string> included_nodes;
  std::unordered_set<std::string> excluded_nodes;
  ResourceView req_node_res_view;
  ResourceView req_task_res_view;
  std::unordered_map<task_id_t, ResourceInNodeV3> task_res_map;             // :760
  std::unordered_map<CranedId, std::set<task_id_t>> craned_task_map;       // :761
  std::string allocated_craneds_regex;
  std::optional<StepInteractiveMeta> ia_meta;
  absl::Time deadline_time;

  ResourceV3 allocated_res;
  std::vector<CranedId> craned_ids;
  std::unordered_set<CranedId> configuring_nodes, execution_nodes;
  absl::Time start_time;
  crane::grpc::JobStatus status{crane::grpc::JobStatus::Pending};

  step_id_t StepId() const { return step_id; }
  void SetAllocatedRes(const ResourceV3& r) { allocated_res = r; }
  const ResourceV3& AllocatedRes() const { return allocated_res; }
  void SetCranedIds(const std::vector<CranedId>& v) { craned_ids = v; }
  const std::vector<CranedId>& CranedIds() const { return craned_ids; }
  void SetConfiguringNodes(const std::unordered_set<CranedId>& n) { configuring_nodes = n; }
  void SetExecutionNodes(const std::unordered_set<CranedId>& n) { execution_nodes = n; }
  void SetStartTime(absl::Time t) { start_time = t; }
  void SetStatus(crane::grpc::JobStatus s) { status = s; }
};

inline CommonStepInCtld* JobInCtld::GetStep(step_id_t step) const {   // CtldPublicDefs.h:1115-1123
  auto it = m_steps_.find(step);
  return it == m_steps_.end() ? nullptr : it->second;
}

}  // namespace Ctld
