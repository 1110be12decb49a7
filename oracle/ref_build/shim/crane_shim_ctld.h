// ORACLE / TEST INFRASTRUCTURE ONLY — part 2 of the environment of the sliced REFERENCE
// sources (see crane_shim.h): what src/CraneCtld/JobScheduler.{h,cpp}'s sliced ranges read
// from the rest of CraneCtld.  Stand-ins written for this build, with the member names the
// slices use; the reference definitions they replace are cited.
#pragma once
#include "crane_shim.h"

// (the PublicHeader slice — ResourceInNodeV3, ResourceV3, ResourceView … — is included
// before this file)

namespace Ctld {

struct CommonStepInCtld;

// src/CraneCtld/CtldPublicDefs.h: JobInCtld, as read by the constructors of
// RnJobInScheduler / PdJobInScheduler (JobScheduler.h:75-90,143-170)
struct JobInCtld {
  job_id_t job_id{0};
  absl::Duration time_limit;
  PartitionId partition_id;
  std::string reservation;
  absl::Time submit_time;
  uint32_t partition_priority{0};
  uint32_t qos_priority{0};
  std::string account;
  std::string qos;
  std::string username;
  std::list<std::string> account_chain;
  absl::Time start_time;
  absl::Time end_time;
  ResourceV3 allocated_res;
  ResourceView allocated_res_view;
  ResourceView req_node_res_view;
  ResourceView req_task_res_view;
  ResourceView req_total_res_view;
  uint32_t node_num{1};
  uint32_t ntasks_per_node_min{1};
  uint32_t ntasks_per_node_max{1};
  uint32_t ntasks{1};
  bool exclusive{false};
  std::unordered_set<std::string> included_nodes;
  std::unordered_set<std::string> excluded_nodes;
  double mandated_priority{0.0};
  crane::grpc::JobToCtld job_to_ctld;

  job_id_t JobId() const { return job_id; }
  absl::Time SubmitTime() const { return submit_time; }
  absl::Time StartTime() const { return start_time; }
  absl::Time EndTime() const { return end_time; }
  const ResourceV3& AllocatedRes() const { return allocated_res; }
  const crane::grpc::JobToCtld& JobToCtld() const { return job_to_ctld; }
  const std::string& Username() const { return username; }

  // step scheduler (CtldPublicDefs.h:899-900,825,1115-1123; SchedulePendingSteps is the sliced definition,
  // CtldPublicDefs.cpp:2038-2159)
  std::queue<step_id_t> pending_step_ids_;
  ResourceV3 step_res_avail_;
  absl::Time deadline_time;
  std::map<step_id_t, CommonStepInCtld*> m_steps_;
  CommonStepInCtld* GetStep(step_id_t step) const;
  uint32_t SchedulePendingSteps(std::vector<CommonStepInCtld*>* scheduled_steps);
};

// src/CraneCtld/CtldPublicDefs.h:92-242 (fields read by the slices)
struct Config {
  struct Priority {
    enum TypeEnum { Basic, MultiFactor };
    TypeEnum Type{Basic};
    bool FavorSmall{true};
    uint64_t MaxAge{7UL * 24 * 3600};
    uint32_t WeightAge{0};
    uint32_t WeightFairShare{0};
    uint32_t WeightJobSize{0};
    uint32_t WeightPartition{0};
    uint32_t WeightQoS{0};
  };
  struct PreemptConf {
    crane::grpc::PreemptType PreemptType{crane::grpc::PreemptType::PREEMPT_NONE};
  };
  Priority PriorityConfig;
  PreemptConf Preempt;
  uint32_t ScheduledBatchSize{100000};
};
inline Config g_config;

// "ExclusivePtr" of util::Synchronized values (crane/Lock.h): pointer-like access.
template <class T>
struct RefPtrLike {
  T* p{nullptr};
  T* operator->() const { return p; }
  T& operator*() const { return *p; }
  explicit operator bool() const { return p != nullptr; }
};
template <class T>
struct RefSynchronized {
  mutable T value;
  RefPtrLike<T> GetExclusivePtr() const { return RefPtrLike<T>{&value}; }
};

// src/CraneCtld/Node/NodeDefs.h:59-123 (fields read by NodeSelect)
struct CranedMeta {
  bool alive{false};
  bool drain{false};
  ResourceInNodeV3 res_total;
};
struct ResvMeta {
  absl::Time start_time;
  absl::Time end_time;
  std::vector<CranedId> craned_ids;
  ResourceV3 res_total;
};
struct PartitionMeta {
  std::vector<CranedId> craned_ids;
};

// Index-ordered association list standing in for the meta container's hash maps: iteration
// is ascending index (the reference's order is unspecified; canonical = ascending index).
template <class V>
class RefAssoc {
  std::vector<std::pair<std::string, V>> v_;
  std::map<std::string, size_t> idx_;

 public:
  using value_type = std::pair<std::string, V>;
  V& add(const std::string& k) {
    idx_[k] = v_.size();
    v_.emplace_back(std::piecewise_construct, std::forward_as_tuple(k), std::forward_as_tuple());
    return v_.back().second;
  }
  auto begin() { return v_.begin(); }
  auto end() { return v_.end(); }
  auto begin() const { return v_.begin(); }
  auto end() const { return v_.end(); }
  auto find(const std::string& k) const {
    auto it = idx_.find(k);
    return it == idx_.end() ? v_.end() : v_.begin() + it->second;
  }
  const V& at(const std::string& k) const { return v_.at(idx_.at(k)).second; }
  V& at(const std::string& k) { return v_.at(idx_.at(k)).second; }
  void reserve(size_t n) { v_.reserve(n); }
  void clear() { v_.clear(); idx_.clear(); }
};

// src/CraneCtld/Node/CranedMetaContainer.h (the three accessors NodeSelect uses,
// JobScheduler.cpp:6569,6585,6628)
struct CranedMetaContainer {
  RefAssoc<RefSynchronized<PartitionMeta>> partitions;
  RefAssoc<RefSynchronized<CranedMeta>> craneds;
  RefAssoc<RefSynchronized<ResvMeta>> reservations;
  RefPtrLike<RefAssoc<RefSynchronized<PartitionMeta>>> GetAllPartitionsMetaMapConstPtr() { return {&partitions}; }
  RefPtrLike<RefAssoc<RefSynchronized<CranedMeta>>> GetCranedMetaMapConstPtr() { return {&craneds}; }
  RefPtrLike<RefAssoc<RefSynchronized<ResvMeta>>> GetResvMetaMapPtr() { return {&reservations}; }
};
inline CranedMetaContainer* g_meta_container = nullptr;

// src/CraneCtld/Account/AccountDefs.h:27-49 (fields read at JobScheduler.cpp:6530-6537 and by the run-limit
// checks, AccountMetaContainer.cpp:508-540,542-670,985-1025)
struct Qos {
  bool deleted{false};
  std::vector<std::string> preempt;
  uint32_t max_jobs_per_user{std::numeric_limits<uint32_t>::max()};
  uint32_t max_jobs_per_account{std::numeric_limits<uint32_t>::max()};
  cpu_t max_cpus_per_user;
  uint32_t max_jobs{std::numeric_limits<uint32_t>::max()};
  absl::Duration max_wall;
  ResourceView max_tres;
  ResourceView max_tres_per_user;
  ResourceView max_tres_per_account;
};
// src/CraneCtld/Account/AccountDefs.h:163-175
struct PartitionResourceLimit {
  ResourceView max_tres;
  ResourceView max_tres_per_job;
  uint32_t max_jobs{std::numeric_limits<uint32_t>::max()};
  uint32_t max_submit_jobs{std::numeric_limits<uint32_t>::max()};
  absl::Duration max_wall{absl::ZeroDuration()};
  absl::Duration max_wall_duration_per_job;
};
using PartitionToLimitMap = std::unordered_map<std::string, PartitionResourceLimit>;   // AccountDefs.h:177-178
// src/CraneCtld/Account/AccountDefs.h:180-207 (fields read at AccountMetaContainer.cpp:958-969)
struct Account {
  std::string name;
  std::string parent_account;
  PartitionToLimitMap partition_to_limit_map;
};
// src/CraneCtld/Account/AccountDefs.h:209-262 (fields read at AccountMetaContainer.cpp:921-933)
struct User {
  struct AttrsInAccount {
    PartitionToLimitMap partition_to_limit_map;
    bool blocked{false};
  };
  using AccountToAttrsMap = std::unordered_map<std::string, AttrsInAccount>;
  std::string name;
  AccountToAttrsMap account_to_attrs_map;
};
// src/CraneCtld/Account/AccountManager.h: GetAllQosInfo (JobScheduler.cpp:6530) and the three look-ups of
// CheckAndMallocMetaResource (AccountMetaContainer.cpp:185,193,201).  The real ones return shared-lock pointers.
struct AccountManager {
  std::map<std::string, std::unique_ptr<Qos>> qos_map;
  std::map<std::string, std::unique_ptr<User>> user_map;
  std::unordered_map<std::string, std::unique_ptr<Account>> account_map;   // AccountMetaContainer::AccountRawMap
  bool have_accounts{true};
  RefPtrLike<std::map<std::string, std::unique_ptr<Qos>>> GetAllQosInfo() { return {&qos_map}; }
  RefPtrLike<const User> GetExistedUserInfo(const std::string& name) {
    auto it = user_map.find(name);
    return {it == user_map.end() ? nullptr : it->second.get()};
  }
  RefPtrLike<const std::unordered_map<std::string, std::unique_ptr<Account>>> GetAllAccountInfo() {
    return {have_accounts ? &account_map : nullptr};
  }
  RefPtrLike<const Qos> GetExistedQosInfo(const std::string& name) {
    auto it = qos_map.find(name);
    return {it == qos_map.end() || it->second->deleted ? nullptr : it->second.get()};
  }
};
inline AccountManager* g_account_manager = nullptr;

struct PdJobInScheduler;
}  // namespace Ctld
#include "license_types.inc"   // struct License, AccountDefs.h (generated at build time)
namespace Ctld {
// src/CraneCtld/Accounting/LicenseManager.h:41-123: the class around the ONE member function the path calls
// (CheckLicenseCountSufficient, compiled from the reference's text: license_impl.inc).  m_licenses_map_ is a
// util::AtomicHashMap<HashMap, LicenseId, License> there (crane/AtomicHashMap.h): GetMapExclusivePtr() dereferences to a map whose
// values hand out the License through GetExclusivePtr() — the stand-in keeps exactly that shape, without the locks.
struct ShimLicenseEntry {
  License value;
  const License* GetExclusivePtr() const { return &value; }
};
struct ShimLicenseMap {
  std::unordered_map<LicenseId, ShimLicenseEntry> map;
  const std::unordered_map<LicenseId, ShimLicenseEntry>* GetMapExclusivePtr() const { return &map; }
};
class LicenseManager {
 public:
  void CheckLicenseCountSufficient(std::vector<PdJobInScheduler*>* job_ptr_vec);
  ShimLicenseMap m_licenses_map_;
};
inline LicenseManager* g_license_manager = nullptr;

// JobScheduler::EnqueuePreemptCancel, JobScheduler.h:1112
struct RefJobSchedulerStub {
  std::vector<job_id_t> cancelled;
  void EnqueuePreemptCancel(std::vector<job_id_t> ids) { cancelled.insert(cancelled.end(), ids.begin(), ids.end()); }
};
inline RefJobSchedulerStub* g_job_scheduler = nullptr;

}  // namespace Ctld
