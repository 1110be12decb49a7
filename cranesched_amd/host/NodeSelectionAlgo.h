// C++ host side of the drop-in: the `INodeSelectionAlgo` plugin surface for CraneCtld's JobScheduler.
//
// The reference holds its node-selection algorithm as a concrete class
//     std::unique_ptr<SchedulerAlgo> m_node_selection_algo_;          src/CraneCtld/JobScheduler.h:1286
// built at src/CraneCtld/JobScheduler.cpp:158-159 and called once per cycle at :1441 through its single
// public method (JobScheduler.h:260-263):
//     void NodeSelect(const absl::Time& now,
//                     const std::vector<std::unique_ptr<RnJobInScheduler>>& running_jobs,
//                     const std::vector<std::unique_ptr<PdJobInScheduler>>& pending_jobs);
// `INodeSelectionAlgo` is that signature as an abstract base; `GpuNodeSelectionAlgo` implements it on top
// of the C ABI in include/crane_gpu/node_select.h, so swapping the make_unique at cpp:158-159 is the whole
// integration.  Struct and field names follow the reference (JobScheduler.h:57-170,
// PublicHeader.h:427-761) so that code written against CraneCtld's types reads the same here.
//
// abseil / protobuf / fpm are not available offline; their types are restated minimally:
//   absl::Time      -> crane::TimeSec   (int64 seconds; the path only uses whole seconds, cpp:1351)
//   cpu_t           -> crane::cpu_t     (int64 raw = value * 256, fpm::fixed<int64,__int128,8>)
//   CranedId/SlotId -> std::string, as in the reference
#pragma once
#include <cstdint>
#include <list>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <variant>
#include <vector>

struct cns_engine;
struct cns_group_info;

namespace crane {

using TimeSec = int64_t;
using job_id_t = uint32_t;
using CranedId = std::string;
using PartitionId = std::string;
using SlotId = std::string;

struct cpu_t {  // fpm::fixed<int64_t, __int128, 8>
  int64_t raw = 0;
  cpu_t() = default;
  explicit cpu_t(int v) : raw(static_cast<int64_t>(v) * 256) {}
  explicit cpu_t(double v) : raw(static_cast<int64_t>(v * 256.0 + (v >= 0 ? 0.5 : -0.5))) {}
  static cpu_t from_raw(int64_t r) { cpu_t c; c.raw = r; return c; }
  explicit operator double() const { return static_cast<double>(raw) / 256.0; }
  bool operator==(const cpu_t& o) const { return raw == o.raw; }
};

struct GresCount {  // PublicHeader.h:505-523
  uint64_t total{0};
  std::unordered_map<std::string /*type*/, uint64_t> specified;
};
using GresMap = std::unordered_map<std::string /*name*/, GresCount>;

struct ResourceView {  // PublicHeader.h:695-761 (fields that reach the path)
  cpu_t cpu_count;
  uint64_t memory_bytes{0};
  uint64_t memory_sw_bytes{0};
  GresMap gres_map;
};

struct CpuSet {  // PublicHeader.h:555-573
  std::set<uint32_t> core_ids;
  cpu_t cpu_count;
};
using TypeSlotsMap = std::unordered_map<std::string /*type*/, std::set<SlotId>>;
using DedicatedResourceInNode = std::unordered_map<std::string /*name*/, TypeSlotsMap>;

struct ResourceInNodeV3 {  // PublicHeader.h:586-639
  CpuSet cpu_set;
  uint64_t memory_bytes{0};
  uint64_t memory_sw_bytes{0};
  DedicatedResourceInNode gres;
};
using ResourceV3 = std::unordered_map<CranedId, ResourceInNodeV3>;  // EachNodeResMap()

struct RnJobInScheduler {  // JobScheduler.h:57-90
  job_id_t job_id{0};
  std::string qos;                    // h:68: read by TryPreempt_ through NodeState::qos_job_map
  PartitionId partition_id;
  std::string reservation;
  TimeSec start_time{0};
  TimeSec end_time{0};
  ResourceV3 allocated_res;
  // read by MultiFactorPriority (JobScheduler.cpp:7692-7746)
  std::string account;
  uint32_t qos_priority{0};
  uint32_t partition_priority{0};
  uint32_t node_num{0};               // uninitialised in the reference (h:70,76-89); here: allocated_res.size() when 0
  ResourceView allocated_res_view;    // cpu_count + memory_bytes of the whole allocation
};

struct PdJobInScheduler;
using PreemptedJob = std::variant<PdJobInScheduler*, RnJobInScheduler*>;   // JobScheduler.h:124-126

struct PdJobInScheduler {  // JobScheduler.h:92-170
  job_id_t job_id{0};
  int64_t time_limit{0};  // absl::Duration, whole seconds
  PartitionId partition_id;
  std::string reservation;
  ResourceView req_node_res_view;
  ResourceView req_task_res_view;
  uint32_t node_num{1};
  uint32_t ntasks_per_node_min{1};
  uint32_t ntasks_per_node_max{1};
  uint32_t ntasks{1};
  bool exclusive{false};
  std::unordered_set<std::string> included_nodes;
  std::unordered_set<std::string> excluded_nodes;
  // results (JobScheduler.h:117-133)
  std::unordered_map<CranedId, uint32_t> craned_id_to_task_num;
  TimeSec start_time{0};
  TimeSec end_time{0};
  ResourceV3 allocated_res;
  std::vector<CranedId> craned_ids;
  std::string reason;
  std::vector<PreemptedJob> preempted_jobs;   // result of TryPreempt_ (JobScheduler.cpp:6490-6496), push_back order
  bool is_scheduled() const { return reason.empty(); }
  // licenses (JobScheduler.h:141-146; LicenseManager::CheckLicenseCountSufficient, LicenseManager.cpp:167-221)
  std::vector<std::pair<std::string, uint32_t>> req_licenses;   // (license id, count), request order
  bool is_license_or{false};
  std::unordered_map<std::string, uint32_t> actual_licenses;    // result
  // read / written by MultiFactorPriority (JobScheduler.cpp:7616, :7664-7690, :7759-7767)
  TimeSec submit_time{0};
  std::string account;
  uint32_t qos_priority{0};
  uint32_t partition_priority{0};
  double priority{0.0};               // cached across cycles: 0.0 = compute (cpp:7616)
  ResourceView req_total_res_view;    // req_node * node_num + req_task * ntasks (cpp:7156)
  // read by the commit loop's run-limit check (AccountMetaContainer::CheckAndMallocMetaResource); the account chain
  // (job.account_chain, JobScheduler.h:108) is derived from AccountMetaSnapshot::account_parent
  std::string username;
  std::string qos;
};

// What NodeSelect's prologue reads from g_meta_container (JobScheduler.cpp:6563-6617): per craned its
// res_total, alive/drain, and the partition -> craned ids map.  In CraneCtld the adapter fills this from
// CranedMetaContainer (src/CraneCtld/Node/CranedMetaContainer.h:116-120, NodeDefs.h:59-81) while holding
// the same locks the reference takes.
struct CranedMeta {
  CranedId craned_id;
  ResourceInNodeV3 res_total;
  bool alive{true};
  bool drain{false};
};
// ResvMeta as NodeSelect reads it (JobScheduler.cpp:6627-6679): window and the reserved resources per node.
struct ResvMeta {
  std::string name;          // ResvId
  TimeSec start_time{0};
  TimeSec end_time{0};
  ResourceV3 res_total;      // EachNodeResMap()
};
struct ClusterSnapshot {
  std::vector<CranedMeta> craned_metas;                                   // dense order = canonical tie-break order
  std::vector<std::pair<PartitionId, std::vector<CranedId>>> partitions;  // PartitionMeta::craned_ids
  std::vector<ResvMeta> reservations;                                     // g_meta_container->GetResvMetaMapPtr(); vector order = canonical order
  // g_config.Preempt.PreemptType != NONE (PREEMPT_QOS): NodeSelect calls LocalScheduler::TryPreempt_ (JobScheduler.cpp:
  // 6140-6143) with the preempt lists of the QoS table (cpp:6532-6543: Qos::preempt of every pending job's qos).  Served by
  // cns_select_preempt (include/crane_gpu/preempt.h) through NodeSelect(now, running_jobs, pending_jobs) — also together
  // with reservations and with partitions that share nodes.  The mirror-fed form NodeSelect(now, pending_jobs) has no
  // RnJobInScheduler objects to hand back in preempted_jobs and is refused for such a snapshot (CNS_ERR_STATE).
  bool preempt_enabled{false};
  std::unordered_map<std::string, std::vector<std::string>> qos_preempt;   // qos name -> Qos::preempt
};

// ---- what the commit loop's run-limit admission reads (JobScheduler.cpp:1557-1573) -----------------------------
// Qos (src/CraneCtld/Account/AccountDefs.h:27-49), the fields CheckRunLimits_ reads
struct Qos {
  uint32_t max_jobs_per_user{UINT32_MAX};
  uint32_t max_jobs_per_account{UINT32_MAX};
  uint32_t max_jobs{UINT32_MAX};
  cpu_t max_cpus_per_user{cpu_t::from_raw(int64_t{1} << 53)};   // kUnlimitedCpu
  int64_t max_wall{0};                                          // absl::Duration seconds, 0 = unlimited
  ResourceView max_tres, max_tres_per_user, max_tres_per_account;
  Qos();
};
struct PartitionResourceLimit {  // AccountDefs.h:163-175
  ResourceView max_tres;
  uint32_t max_jobs{UINT32_MAX};
  int64_t max_wall{0};
  PartitionResourceLimit();
};
struct MetaResource {  // AccountMetaContainer.h:30-35
  ResourceView resource;
  uint32_t jobs_count{0};
  uint32_t submit_jobs_count{0};   // counted at submit time (h:33); the run-limit admission neither reads nor changes it
  int64_t wall_time{0};
};
struct MetaResourceStat {  // AccountMetaContainer.h:62-80
  std::unordered_map<std::string /*qos*/, MetaResource> qos_to_resource_map;
  std::unordered_map<std::string /*account*/, std::unordered_map<PartitionId, MetaResource>> account_to_partition_to_resource_map;  // users
  std::unordered_map<PartitionId, MetaResource> partition_to_resource_map;                                                          // accounts
};
// The state of AccountManager + AccountMetaContainer the check reads, copied by the caller under the locks the
// reference takes (AccountMetaContainer.cpp:204-207).
struct AccountMetaSnapshot {
  std::unordered_map<std::string, Qos> qos;                                              // GetExistedQosInfo
  std::unordered_map<std::string, std::string> account_parent;                           // Account::parent_account, "" = root
  std::unordered_map<std::string, std::unordered_map<PartitionId, PartitionResourceLimit>> account_partition_limits;  // Account::partition_to_limit_map
  // User::account_to_attrs_map: the accounts of a user, each with its partition_to_limit_map
  std::unordered_map<std::string, std::unordered_map<std::string, std::unordered_map<PartitionId, PartitionResourceLimit>>> user_accounts;
  std::unordered_map<std::string, MetaResourceStat> user_meta;     // m_user_meta_map_
  std::unordered_map<std::string, MetaResourceStat> account_meta;  // m_account_meta_map_
  std::unordered_map<std::string, MetaResource> qos_meta;          // m_qos_meta_map_
};

// ---- step scheduling (JobInCtld::SchedulePendingSteps, CtldPublicDefs.cpp:2038-2159) ---------------------------
// The CommonStepInCtld fields that function reads and writes.
struct StepInScheduler {
  uint32_t step_id{0};
  ResourceView req_node_res_view, req_task_res_view;
  uint32_t node_num{1}, ntasks{1}, ntasks_per_node_min{1}, ntasks_per_node_max{1};
  std::unordered_set<std::string> included_nodes, excluded_nodes;
  // results (:2129-2135)
  bool scheduled{false};
  std::vector<CranedId> craned_ids;                                   // in the order the nodes were handed tasks
  ResourceV3 allocated_res;                                           // step_alloc_res
  std::unordered_map<CranedId, std::set<uint32_t>> craned_task_map;   // task ids per node
  std::unordered_map<uint32_t, ResourceInNodeV3> task_res_map;        // per task id
};
// One running job with pending steps: its step_res_avail_ (updated in place) and pending_step_ids_ in queue order.
struct JobStepQueue {
  job_id_t job_id{0};
  ResourceV3* step_res_avail{nullptr};
  std::vector<StepInScheduler*> pending_steps;
};

// License (LicenseManager: total, used, reserved, last_deficit as read at LicenseManager.cpp:188-189,203-204)
struct License {
  uint32_t total{0}, used{0}, reserved{0}, last_deficit{0};
};

// g_config.PriorityConfig, CtldPublicDefs.h:162-174
struct PriorityConfig {
  bool FavorSmall{true};
  uint64_t MaxAge{14 * 24 * 3600};
  uint32_t WeightAge{500}, WeightFairShare{10000}, WeightJobSize{0}, WeightPartition{1000}, WeightQoS{1000000};
};

// JobScheduler.h:174-181: what NodeSelect calls at JobScheduler.cpp:6735 before its ordered loop.
class IPrioritySorter {
 public:
  virtual ~IPrioritySorter() = default;
  virtual void GetOrderedJobPtrVec(const TimeSec& now, const std::vector<std::unique_ptr<PdJobInScheduler>>& pending_jobs,
                                   const std::vector<std::unique_ptr<RnJobInScheduler>>& running_jobs, size_t limit,
                                   std::vector<PdJobInScheduler*>& job_ptr_vec) = 0;
};

// MultiFactorPriority (JobScheduler.h:203-231, JobScheduler.cpp:7606-7819) on the MI355X: bounds, service
// values, priorities and the ordering run on the device (include/crane_gpu/priority.h).  Writes job->priority,
// fills job_ptr_vec with the first `limit` jobs by descending priority (ties: input order) and marks the rest
// "Priority" (cpp:7625-7630).  On an engine error the vector keeps input order and LastError() says why.
class GpuMultiFactorPriority final : public IPrioritySorter {
 public:
  explicit GpuMultiFactorPriority(const PriorityConfig& cfg, int device = 0);
  ~GpuMultiFactorPriority() override;
  GpuMultiFactorPriority(const GpuMultiFactorPriority&) = delete;
  GpuMultiFactorPriority& operator=(const GpuMultiFactorPriority&) = delete;
  void GetOrderedJobPtrVec(const TimeSec& now, const std::vector<std::unique_ptr<PdJobInScheduler>>& pending_jobs,
                           const std::vector<std::unique_ptr<RnJobInScheduler>>& running_jobs, size_t limit,
                           std::vector<PdJobInScheduler*>& job_ptr_vec) override;
  bool Ok() const { return status_ == 0; }
  const std::string& LastError() const { return error_; }

 private:
  PriorityConfig cfg_;
  cns_engine* h_{nullptr};
  int status_{0};
  std::string error_;
};

class INodeSelectionAlgo {
 public:
  virtual ~INodeSelectionAlgo() = default;
  virtual void NodeSelect(const TimeSec& now,
                          const std::vector<std::unique_ptr<RnJobInScheduler>>& running_jobs,
                          const std::vector<std::unique_ptr<PdJobInScheduler>>& pending_jobs) = 0;
};

// MI355X implementation.  It has no CPU path of its own: on an engine-level failure (no GPU, unsupported input) every pending job
// is left unscheduled with reason "GpuEngineError", Ok() / LastStatus() / LastError() say why, and the CALLER decides — INTEGRATION.md
// §3 wires the reference's own SchedulerAlgo behind this class for exactly that case (two lines in JobScheduler's constructor).
class GpuNodeSelectionAlgo final : public INodeSelectionAlgo {
 public:
  explicit GpuNodeSelectionAlgo(int device = 0, uint64_t scheduled_batch_size = 0);
  // Several devices of one node: the groups of partitions connected through shared nodes are dealt over them (group g -> devices[g % N]),
  // every device runs its shard of the ordered queue on its own host thread, ONE all-gather (RCCL) merges the packed results
  // (include/crane_gpu/node_select.h, "several devices").  {d} is the one-device form.
  explicit GpuNodeSelectionAlgo(const std::vector<int>& devices, uint64_t scheduled_batch_size = 0);
  // What lies outside the engine's limits (a node with a core id >= 256, GRES classes / slots beyond the 64-bit mask, the 65th distinct
  // res_total record, a group of partitions wider than the widest tile — the reference bounds none of it, PublicHeader.h:555-573,427-494)
  // refuses ONLY the group of partitions it touches: their jobs leave NodeSelect with reason "GpuEngineRefused" and nothing else set, every
  // other partition is served.  RefusedJobs(): exactly those jobs of the last cycle, in its order — the caller's CPU SchedulerAlgo takes
  // them.  RefusedPartitions(): which partitions of the snapshot, and UnsupportedNodes(): how many nodes caused it.
  const std::vector<const PdJobInScheduler*>& RefusedJobs() const;
  std::vector<PartitionId> RefusedPartitions() const;
  size_t UnsupportedNodes() const;
  size_t NumDevices() const;
  bool LastGroupInfo(cns_group_info* out) const;   // false on one device
  ~GpuNodeSelectionAlgo() override;
  GpuNodeSelectionAlgo(const GpuNodeSelectionAlgo&) = delete;
  GpuNodeSelectionAlgo& operator=(const GpuNodeSelectionAlgo&) = delete;

  // Per-cycle snapshot (the reference re-reads the meta container inside NodeSelect; here the caller
  // hands the snapshot over before the call).
  void SetClusterSnapshot(const ClusterSnapshot& snap);
  // Incremental form for what changes between cycles without changing the node set (CranedMetaContainer::CranedUp /
  // CranedDown, CranedMetaContainer.cpp:26-122, and the drain flag): flips the node's "schedulable" bit
  // (alive && !drain, JobScheduler.cpp:6595) and re-sends the packed tables; the snapshot's dense indices, the
  // reservation tables and the cached running allocations stay valid.  SetClusterSnapshot is only needed again when
  // nodes, partitions, res_total or reservations change.
  void SetCranedState(const CranedId& craned_id, bool alive, bool drain);
  // Optional: the sorter NodeSelect consults first (SchedulerAlgo's ctor argument, JobScheduler.h:247);
  // nullptr = BasicPriority (input order, JobScheduler.h:185-200).  Not owned.
  void SetPrioritySorter(IPrioritySorter* sorter) { sorter_ = sorter; }
  // Default: jobs that did not start now (pending reason set) get reason, start_time and end_time only — all the commit
  // loop reads of them (JobScheduler.cpp:1503-1510).  SetFullWriteBack(true) also fills craned_ids / allocated_res of the
  // jobs NodeSelect backfilled for later, exactly as the reference's NodeSelect leaves them.
  void SetFullWriteBack(bool full);
  // Deferred write-back: a job that starts now leaves NodeSelect with reason, start_time, end_time and craned_ids — everything the
  // commit loop reads up to the run-limit admission (JobScheduler.cpp:1503-1573); MaterializeAllocation(job) then fills its
  // craned_id_to_task_num / allocated_res (exactly what the full write-back builds) for the jobs that ARE launched (:1590-1600),
  // from the packed placements the adapter keeps.  0.39 -> 0.1 us per started job in NodeSelect; false: the job was not placed.
  void SetDeferredWriteBack(bool deferred);
  // Host threads for the two per-job loops of a cycle (packing the pending jobs, the write-back): default 1, like the reference's
  // single ScheduleThread; jobs are independent there, so n threads take n slices of the ordered vector.  The engine's own pass over the
  // queue inside cns_select follows the same number (cns_set_host_threads on every device's engine); without a call it keeps its
  // default (CNS_HOST_THREADS from the environment, else up to 16 threads for queues of 32 768 jobs and more).
  void SetHostThreads(int n);
  bool MaterializeAllocation(PdJobInScheduler& job);
  // The cycle's license table for the pre-pass NodeSelect runs between ordering and selection
  // (g_license_manager->CheckLicenseCountSufficient, JobScheduler.cpp:6739; LicenseManager.cpp:167-221): a tiny
  // sequential counter pass over the ordered jobs, done on the host; jobs it rejects get reason "License" and are
  // not given to the device.  Empty table + no requests = no-op.
  void SetLicenses(std::unordered_map<std::string, License> licenses) { licenses_ = std::move(licenses); }
  // ... the pass itself (host-only; NodeSelect calls it with the table of SetLicenses and the sorter's order)
  static void CheckLicenseCountSufficient(const std::unordered_map<std::string, License>& licenses, const std::vector<PdJobInScheduler*>& ord);

  void NodeSelect(const TimeSec& now, const std::vector<std::unique_ptr<RnJobInScheduler>>& running_jobs,
                  const std::vector<std::unique_ptr<PdJobInScheduler>>& pending_jobs) override;

  // ---- event-fed mirror of the running allocations -------------------------------------------------
  // Instead of re-deriving the running jobs' allocations from the vector NodeSelect is handed every cycle, the adapter
  // can be told what the meta container is told: the same calls, at the same places (JobScheduler.cpp:1590-1612 for the
  // start of a job, the job-end path for the release) —
  //   MallocResourceFromNode(craned, job, resources)   CranedMetaContainer.cpp:178-224
  //   FreeResourceFromNode(craned, job)                CranedMetaContainer.cpp:226-277
  // plus the two facts of RnJobInScheduler the meta container does not hold: the job's end time and its reservation.
  // The allocation is packed ONCE, when it is made; NodeSelect(now, pending_jobs) then runs the cycle on the mirror
  // (running jobs in ascending job id, the order of the reference's running-job map).  A new snapshot re-packs the
  // mirror (dense node indices and GRES bit positions are per snapshot).
  void MallocResourceFromNode(const CranedId& craned_id, job_id_t job_id, const ResourceV3& resources);
  void FreeResourceFromNode(const CranedId& craned_id, job_id_t job_id);
  void SetRunningJobInfo(job_id_t job_id, TimeSec end_time, const std::string& reservation = "");
  size_t MirroredRunningJobs() const;
  // how the mirror was packed so far: full walks over the job map / patches of the packed form kept from the previous cycle
  void MirrorPackCounts(size_t* full_walks, size_t* patches) const;
  // (`running_for_priority`: only read by a multifactor sorter, JobScheduler.cpp:7692-7746; not needed with BasicPriority)
  void NodeSelect(const TimeSec& now, const std::vector<std::unique_ptr<PdJobInScheduler>>& pending_jobs,
                  const std::vector<std::unique_ptr<RnJobInScheduler>>* running_for_priority = nullptr);

  // The run-limit admission of the commit loop, batched: AccountMetaContainer::CheckAndMallocMetaResource
  // (AccountMetaContainer.cpp:180-224) for every job of `pending_jobs` — in THAT order, JobScheduler.cpp:1492 — that the
  // last NodeSelect started (`reason` empty).  results[i] = "" (admitted: `meta`'s usage maps now include the job, as
  // after DoMallocResource_) or the pending reason the reference would set ("QosJobsResourceLimit", ...); jobs with a
  // reason keep it.  The NodeSelect results are read where they are, on the device (include/crane_gpu/run_limits.h).
  // GRES names / types no node of the snapshot has are dropped from limits and usage (no job can allocate them).
  void CheckAndMallocMetaResource(AccountMetaSnapshot& meta, const std::vector<std::unique_ptr<PdJobInScheduler>>& pending_jobs,
                                  std::vector<std::string>& results);

  // JobScheduler::StepScheduleThread_'s loop (JobScheduler.cpp:1992-2001) in one call: SchedulePendingSteps for every
  // job of `jobs`.  Steps that were scheduled get `scheduled`, craned_ids, allocated_res, craned_task_map and
  // task_res_map; each job's step_res_avail is updated; the first step of a job that does not fit stops that job's
  // queue.  A job's nodes are walked in the order of the snapshot (the reference walks an unordered_map).
  void SchedulePendingSteps(std::vector<JobStepQueue>& jobs);

  // Measurement / test hook: only the host-side packing of the running jobs (what NodeSelect does before
  // cns_set_running), with or without the per-job cache; needs no device.  Returns the number of allocation records,
  // *checksum covers every array cns_set_running would receive, *pack_ms is the packing alone.
  // ... and of the pending side: cns_job_soa packing + the write-back of (synthetic) placements into the jobs.
  void PendingCycleForBench(const std::vector<std::unique_ptr<PdJobInScheduler>>& pending_jobs, double* pack_ms,
                            double* write_back_ms, uint64_t* checksum);
  size_t PackRunningForBench(const std::vector<std::unique_ptr<RnJobInScheduler>>& running_jobs, bool use_cache,
                             uint64_t* checksum, double* pack_ms = nullptr);
  // ... the same from the event-fed mirror.  *checksum_canonical (both functions) does not depend on the order of a job's
  // per-node records (the explicit path walks an unordered_map, the mirror keeps event order).
  size_t PackMirrorForBench(uint64_t* checksum_canonical, double* pack_ms = nullptr);
  uint64_t LastRunningChecksumCanonical() const;
  // true: the running set packed last (PackRunningForBench / PackMirrorForBench / a NodeSelect) holds an allocation with a core
  // id >= 256 — such a cycle is refused with CNS_ERR_UNSUPPORTED for as long as the job runs (the bit is kept with the cached /
  // mirrored record, not with the cycle that first packed it)
  bool PackedRunningSetOverflows() const;

  // ---- placement -> wire: the protobuf encoding of what JobToD carries, written straight from the
  // packed placements of the last NodeSelect — no ResourceInNodeV3 object, no std::set, no map is built on the way.
  //   crane.grpc.ResourceInNodeV3 {cpu_ids=1 (packed), cpu_count=2, memory_bytes=3, memory_sw_bytes=4, gres=5}
  //                                                  protos/PublicDefs.proto:63-69, PublicHeader.cpp:981-998,255-266
  //   crane.grpc.JobToD {job_id=1, uid=2, res=4, partition=5, account=6, qos=7, name=9}
  //                                                  protos/PublicDefs.proto:396-409, CtldPublicDefs.cpp:537-554
  // Fields are written in field-number order, map entries in (name, type) order and slots in std::set order, i.e. what
  // protobuf's deterministic serialisation of the reference's message produces.
  struct WireBatch {
    struct Rec { uint32_t job, node, off, len; };   // job: index in the vector handed to NodeSelect's order (LastOrder())
    std::string bytes;                              // the serialised ResourceInNodeV3 messages, back to back
    std::vector<Rec> recs;                          // one per (job that starts now, allocated node), queue order
  };
  // every allocation of the jobs that start in this cycle; returns the number of records
  size_t EmitStartedResourcesWire(WireBatch* out) const;
  const std::vector<const PdJobInScheduler*>& LastOrder() const;
  // one allocation / one JobToD of a job of the last cycle (appends to *out; false: job or node not in the last cycle's result)
  bool AppendResourceInNodeV3Wire(const PdJobInScheduler& job, const CranedId& craned_id, std::string* out);
  // JobToD.array_task (field 16, ArrayTaskIdentity{array_job_id, task_id}): set by the reference for array children only
  // (GetArrayTaskIdentity(), CtldPublicDefs.cpp:547-551); pass the identity to emit it, nullptr for every other job.
  struct ArrayTaskIdentity { uint32_t array_job_id; uint32_t task_id; };
  bool AppendJobToDWire(const PdJobInScheduler& job, uint32_t uid, const std::string& name, const CranedId& craned_id,
                        std::string* out, const ArrayTaskIdentity* array_task = nullptr);
  static void ComposeJobToDWire(uint32_t job_id, uint32_t uid, const std::string& partition, const std::string& account,
                                const std::string& qos, const std::string& name, const std::string& res_wire, std::string* out,
                                const ArrayTaskIdentity* array_task = nullptr);
  // test / bench hook (no device): the wire bytes and the ResourceInNodeV3 object of one packed allocation
  void WireOfPackedForTest(int64_t cpu_raw, uint64_t mem, uint64_t mem_sw, uint64_t core_lo, uint64_t core_hi, uint64_t gres,
                           std::string* wire, ResourceInNodeV3* obj, uint64_t core_w2 = 0, uint64_t core_w3 = 0) const;
  // ... and the emission of PendingCycleForBench's synthetic placements (ms per call)
  double EmitWireForBench(size_t* records, size_t* bytes);

  // ---- preemption (only when the snapshot had preempt_enabled) -------------------------------------------------------
  // m_preempting_set_ (JobScheduler.h:984) lives in the adapter across cycles; the running jobs newly put into it by the
  // last cycle are what the reference hands to EnqueuePreemptCancel (JobScheduler.cpp:6793), in that order.
  const std::vector<job_id_t>& LastPreemptCancel() const;
  const std::set<job_id_t>& PreemptingSet() const;

  bool Ok() const { return status_ == 0; }
  // wall time of the three parts of the last NodeSelect: packing the pending jobs | cns_select (H2D of the job table, the kernels, D2H
  // of the placements; the arrays live in page-locked memory of the engine, kept across cycles) | write-back into the job objects
  void LastCycleMs(double* pack_ms, double* engine_ms, double* write_back_ms) const;
  int LastStatus() const { return status_; }
  const std::string& LastError() const { return error_; }

 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
  void SelectPacked_(const TimeSec& now, const std::vector<std::unique_ptr<PdJobInScheduler>>& pending_jobs,
                     const std::vector<std::unique_ptr<RnJobInScheduler>>& running_jobs);
  IPrioritySorter* sorter_{nullptr};
  std::unordered_map<std::string, License> licenses_;
  uint64_t batch_{0};
  int status_{0};
  std::string error_;
};

}  // namespace crane
