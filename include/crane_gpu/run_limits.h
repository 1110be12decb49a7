/* C ABI of the MI355X run-limit admission pass — the QoS / account / partition "post-filter" that
 * JobScheduler::ScheduleThread_ applies to the jobs NodeSelect placed at `now` (SURVEY.md §8(f)-1, and the
 * "+ per-account/QoS limits" of benchmark config C4).
 *
 * Reference (paths relative to the CraneSched tree):
 *   call site, commit loop in pending-vector order          src/CraneCtld/JobScheduler.cpp:1492-1573
 *     (only jobs whose NodeSelect reason is "" get here: a non-empty reason `continue`s at :1507-1510)
 *   AccountMetaContainer::CheckAndMallocMetaResource        src/CraneCtld/Accounting/AccountMetaContainer.cpp:180-224
 *   CheckRunLimits_  (user -> account chain -> global QoS)  :891-1028
 *   CheckQosRunLimitsForEntity_ / CheckPartitionRunLimitsForEntity_ / CheckEntityRunLimits_   :508-688
 *   CheckTres_ / IsUnlimitedTres_ / CheckGres_              :345-365,1030-1050
 *   DoMallocResource_  (the counters a started job adds to) :1067-1124
 *   Qos / PartitionResourceLimit                            src/CraneCtld/Account/AccountDefs.h:27-49,163-175
 *   MetaResource / MetaResourceStat                         src/CraneCtld/Accounting/AccountMetaContainer.h:30-80
 *
 * What it computes: a greedy admission in pending-vector order.  A candidate is admitted when, for every entity
 * it belongs to — its user (per QoS, and per (account, partition)), every account of its account chain (per QoS,
 * per partition) and the QoS globally — usage + the job's allocation stays within the limits; an admitted job's
 * allocation, 1 job and its time limit are then added to all those usage records (later jobs see them).  A
 * rejected job keeps the nodes NodeSelect reserved for it inside the cycle's model (nothing is returned).
 *
 * Canonical integer model (extends node_select.h's):
 *   user / account / qos / partition : dense indices.  (user, account) pairs that exist in
 *            User::account_to_attrs_map are dense "user_acct" indices.
 *   usage records are dense tables: user_qos[user*Q + qos], user_part[user_acct*Pn + partition],
 *            acct_qos[account*Q + qos], acct_part[account*Pn + partition], qos_usage[qos]; an `exists` byte per
 *            record mirrors "the map has an entry" (QosEntryNotFound / PartitionEntryNotFound).
 *   a ResourceView (allocation, usage or limit) = cpu raw, mem bytes, and per GRES name / (name,type) class of the
 *            handle's cns_gres_layout a count.  A usage / allocation GresMap holds an entry exactly for the counts
 *            that are > 0 (the reference erases zero entries, PublicHeader.cpp:441,468); a LIMIT says which entries
 *            it holds with name_mask / class_mask.
 *   CheckGres_ walks unordered_maps (name order and type order unspecified) and RETURNS TRUE at the first
 *            requested name / type the limit has no entry for (:1034,1043).  Canonical order here: ascending name
 *            index, the name's total first, then its classes in ascending class index.
 *   The account chain is walked from the job's account to the root (job.account_chain order), at most
 *            CNS_LIM_MAX_CHAIN accounts.
 *
 * No CPU fallback: CNS_ERR_NO_DEVICE without a GPU.  One caller thread per handle.
 */
#ifndef CRANE_GPU_RUN_LIMITS_H_
#define CRANE_GPU_RUN_LIMITS_H_

#include <stdint.h>

#include "node_select.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CNS_LIM_NONE 0xFFFFFFFFu
#define CNS_LIM_MAX_CHAIN 6u            /* accounts between a job's account and the root, inclusive */
#define CNS_LIM_UNLIMITED_JOBS 0xFFFFFFFFu /* std::numeric_limits<uint32_t>::max() */
#define CNS_LIM_UNLIMITED_CPU_RAW (INT64_C(1) << 53)       /* kUnlimitedCpu, PublicHeader.h:51 */
#define CNS_LIM_MAX_JOB_MEMORY UINT64_C(10737418240000)    /* kMaxJobMemoryBytes, PublicHeader.h:165 */

/* job.pending_reason strings of the admission (AccountMetaContainer.cpp, lines in brackets) */
typedef enum cns_limit_reason {
  CNS_LIM_ADMITTED = 0,
  CNS_LIM_QOS_ENTRY_NOT_FOUND = 1,        /* "QosEntryNotFound"            [:516] */
  CNS_LIM_QOS_CPU = 2,                    /* "QosCpuResourceLimit"         [:525] */
  CNS_LIM_QOS_JOBS = 3,                   /* "QosJobsResourceLimit"        [:527,534,1001] */
  CNS_LIM_QOS_WALL = 4,                   /* "QosWallTimeLimit"            [:530,537,1012] */
  CNS_LIM_CPU = 5,                        /* "QosCpuResourceLimit"         [:349; CheckTres_'s prefix defaults to "Qos", .h:179-181] */
  CNS_LIM_MEM = 6,                        /* "QosMemResourceLimit"         [:353] */
  CNS_LIM_GRES = 7,                       /* "QosGresResourceLimit"        [:357] */
  CNS_LIM_PARTITION_ENTRY_NOT_FOUND = 8,  /* "PartitionEntryNotFound"      [:559,568,621] */
  CNS_LIM_USER_PARTITION_JOBS = 9,        /* "UserPartitionJobsLimit"      [:580] */
  CNS_LIM_USER_PARTITION_WALL = 10,       /* "UserPartitionWallTimeLimit"  [:594] */
  CNS_LIM_ACC_PARTITION_JOBS = 11,        /* "AccPartitionJobsLimit"       [:633] */
  CNS_LIM_ACC_PARTITION_WALL = 12,        /* "AccPartitionWallTimeLimit"   [:647] */
  CNS_LIM_PARTITION_CPU = 13,             /* "PartitionCpuResourceLimit"   [:602,655 -> :349] */
  CNS_LIM_PARTITION_MEM = 14,             /* "PartitionMemResourceLimit"   */
  CNS_LIM_PARTITION_GRES = 15,            /* "PartitionGresResourceLimit"  */
  CNS_LIM_NOT_CANDIDATE = 255             /* NodeSelect left a pending reason, or the caller set `skip` */
} cns_limit_reason;

/* One ResourceView used as a LIMIT (Qos::max_tres*, PartitionResourceLimit::max_tres). */
typedef struct cns_tres {
  int64_t cpu_raw;
  uint64_t mem;
  uint32_t name_mask;                           /* bit n: the GresMap has an entry for name n      */
  uint32_t class_mask;                          /* bit g: that entry's `specified` has class g     */
  uint64_t name_total[CNS_MAX_GRES_NAMES];      /* GresCount.total                                 */
  uint64_t class_count[CNS_MAX_GRES_CLASSES];   /* GresCount.specified[type]                       */
} cns_tres;

/* The fields of Qos the run checks read (AccountDefs.h:33-45). */
typedef struct cns_qos_limits {
  uint32_t max_jobs_per_user;
  uint32_t max_jobs_per_account;
  uint32_t max_jobs;
  uint32_t reserved0;
  int64_t max_cpus_per_user_raw;
  int64_t max_wall_sec;                 /* 0 = unlimited (absl::ZeroDuration) */
  cns_tres max_tres;
  cns_tres max_tres_per_user;
  cns_tres max_tres_per_account;
} cns_qos_limits;

/* PartitionResourceLimit (AccountDefs.h:163-175), the fields the run checks read. */
typedef struct cns_part_limit {
  uint32_t max_jobs;                    /* CNS_LIM_UNLIMITED_JOBS = unlimited */
  uint32_t reserved0;
  int64_t max_wall_sec;                 /* 0 = unlimited */
  cns_tres max_tres;
} cns_part_limit;

/* One MetaResource (AccountMetaContainer.h:30-35; submit_jobs_count is not touched by this path). */
typedef struct cns_usage {
  int64_t cpu_raw;
  uint64_t mem;
  int64_t wall_sec;
  uint32_t jobs_count;
  uint32_t reserved0;
  uint64_t name_total[CNS_MAX_GRES_NAMES];
  uint64_t class_count[CNS_MAX_GRES_CLASSES];
} cns_usage;

/* Limits + usage at the start of the commit loop.  Everything is copied. */
typedef struct cns_limit_tables {
  uint32_t num_users, num_user_accts, num_accounts, num_qos, num_partitions, num_part_limits;
  const cns_qos_limits* qos;            /* [num_qos]                                                        */
  const uint32_t* acct_parent;          /* [num_accounts] parent account or CNS_LIM_NONE (root)             */
  const cns_part_limit* part_limits;    /* [num_part_limits]                                                */
  const uint32_t* user_part_limit;      /* [num_user_accts*num_partitions] index into part_limits or CNS_LIM_NONE
                                           (User::account_to_attrs_map[acct].partition_to_limit_map); NULL = none */
  const uint32_t* acct_part_limit;      /* [num_accounts*num_partitions] (Account::partition_to_limit_map); NULL = none */
  const cns_usage* user_qos;            /* [num_users*num_qos]   m_user_meta_map_[u].qos_to_resource_map; NULL = 0  */
  const uint8_t* user_qos_exists;       /* NULL = every entry exists                                        */
  const cns_usage* user_part;           /* [num_user_accts*num_partitions] ...account_to_partition_to_resource_map  */
  const uint8_t* user_part_exists;
  const cns_usage* acct_qos;            /* [num_accounts*num_qos] m_account_meta_map_[a].qos_to_resource_map        */
  const uint8_t* acct_qos_exists;
  const cns_usage* acct_part;           /* [num_accounts*num_partitions] ...partition_to_resource_map       */
  const uint8_t* acct_part_exists;
  const cns_usage* qos_usage;           /* [num_qos] m_qos_meta_map_                                        */
} cns_limit_tables;

/* The pending vector of the commit loop (JobScheduler.cpp:1492), in ITS order (ascending job id), which is not
 * the priority order NodeSelect ran in when a multifactor sorter is used. */
typedef struct cns_limit_job_soa {
  uint64_t num_jobs;
  const uint64_t* select_index;   /* [J] index of the job in the cns_job_soa of the last cns_select; NULL = identity */
  const uint32_t* user;           /* [J] < num_users                                                        */
  const uint32_t* user_acct;      /* [J] < num_user_accts : (job.username, job.account)                     */
  const uint32_t* account;        /* [J] < num_accounts   : job.account_chain.front()                       */
  const uint32_t* qos;            /* [J] < num_qos                                                          */
  const uint32_t* partition;      /* [J] < num_partitions : job.partition_id                                */
  const int64_t* time_limit_sec;  /* [J] job.time_limit                                                     */
  const uint8_t* skip;            /* [J] non-zero: the commit loop `continue`d before the check (:1511-1563) or a
                                     name lookup failed in the adapter; NULL = 0                            */
} cns_limit_job_soa;

typedef struct cns_limit_timing {
  double h2d_ms;      /* key upload                                                     */
  double prep_ms;     /* allocation views + key records (job-parallel kernel)           */
  double admit_ms;    /* the ordered admission                                          */
  double d2h_ms;
  uint64_t candidates; /* jobs that reached CheckAndMallocMetaResource                  */
  uint64_t admitted;
  uint32_t rounds;            /* bracketing rounds of the parallel pass (0: not used)    */
  uint32_t ordered_fallback;  /* 1: the ordered single-wave kernel decided (CNS_LIMITS_MODE=seq, or the rounds did not converge) */
} cns_limit_timing;

int cns_set_run_limits(cns_handle* h, const cns_limit_tables* t);

/* The admission over the results of the last cns_select / cns_run_resident, which stay on the device.
 * limit_reason_out[i] (cns_limit_reason) for the i-th job of `jobs`. */
int cns_apply_run_limits(cns_handle* h, const cns_limit_job_soa* jobs, uint8_t* limit_reason_out,
                         uint64_t* num_admitted);

/* Split form (benchmark: keys resident before the timed region). */
int cns_upload_limit_jobs(cns_handle* h, const cns_limit_job_soa* jobs);
int cns_run_limits_resident(cns_handle* h);   /* re-runs from the usage given to cns_set_run_limits */
int cns_download_limits(cns_handle* h, uint8_t* limit_reason_out, uint64_t* num_admitted);

int cns_get_limit_timing(const cns_handle* h, cns_limit_timing* t);

/* Parity / write-back: usage tables after the last admission (what DoMallocResource_ left), same shapes as in
 * cns_limit_tables; any pointer may be NULL. */
int cns_get_usage(cns_handle* h, cns_usage* user_qos, uint8_t* user_qos_exists, cns_usage* user_part,
                  uint8_t* user_part_exists, cns_usage* acct_qos, uint8_t* acct_qos_exists, cns_usage* acct_part,
                  uint8_t* acct_part_exists, cns_usage* qos_usage);

#ifdef __cplusplus
}
#endif
#endif /* CRANE_GPU_RUN_LIMITS_H_ */
