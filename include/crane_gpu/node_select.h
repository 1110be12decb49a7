/*
 * crane_gpu/node_select.h — C ABI of the MI355X node-selection engine.
 *
 * This is the drop-in boundary for CraneCtld's pending-job x node matching
 * hot path.  Everything here is plain C: pointers, sizes, fixed-width ints.
 * No torch / HIP types appear in any signature (streams are `void*`).
 *
 * What it replaces in the reference (paths relative to the CraneSched tree):
 *   - SchedulerAlgo::NodeSelect            src/CraneCtld/JobScheduler.cpp:6507-6836
 *     (declared src/CraneCtld/JobScheduler.h:260-263, called once per cycle
 *      from JobScheduler::ScheduleThread_, JobScheduler.cpp:1441)
 *   - LocalScheduler::GetNodesAndTrySchedule_ / Backfill_
 *                                          src/CraneCtld/JobScheduler.cpp:6147-6376
 *   - NodeState / NodeSelector / MinCpuTimeRatioFirst
 *                                          src/CraneCtld/JobScheduler.h:41-55,272-595
 *   - ResourceView::GetFeasibleResourceInNode, ResourceInNodeV3::Ckmin, <=, +=, -=
 *                                          src/Utilities/PublicHeader/PublicHeader.cpp:519-599,781-827,886-890
 *
 * The reference has no FFI for this path (SchedulerAlgo is a concrete C++
 * class); the C++ adapter in cranesched_amd/host/ exposes `INodeSelectionAlgo`
 * with the reference's NodeSelect signature and calls these entry points.
 * INTEGRATION.md shows the binding a CraneCtld maintainer would add.
 *
 * Canonical integer model:
 *   cpu    : int64 raw fixed point, value*256   (cpu_t = fpm::fixed<int64,__int128,8>,
 *            src/Utilities/PublicHeader/include/crane/PublicHeader.h:44)
 *   mem    : uint64 bytes (mem_sw is never tested on this path and is not carried)
 *   cores  : 256-bit mask over core ids 0..255 (core_lo = ids 0..63, core_hi = 64..127, core_w2 = 128..191,
 *            core_w3 = 192..255; the last two are optional planes at the END of every struct that carries core ids:
 *            NULL = no core id above 127, the layout of ABI 2 is a prefix of ABI 3)
 *   gres   : 64-bit slot mask; class g = (name,type) owns bits
 *            [class_shift[g], class_shift[g]+class_width[g]); bit order inside a
 *            class = lexicographic order of the slot's device path (the order of
 *            std::set<SlotId>, PublicHeader.h:425-428)
 *   time   : int64 seconds; "now" is whole seconds (JobScheduler.cpp:1351)
 *   node   : dense index 0..num_nodes-1; cost ties break on the ascending dense
 *            index (canonical replacement for the reference's pointer order,
 *            JobScheduler.h:594)
 *
 * Threading: one caller thread per handle, not re-entrant (matches
 * ScheduleThread, JobScheduler.cpp:1322).  All functions return 0 on success
 * and a negative cns_status on error; they never throw and never abort.
 * There is NO CPU fallback: if no HIP device is usable cns_create fails.
 */
#ifndef CRANE_GPU_NODE_SELECT_H_
#define CRANE_GPU_NODE_SELECT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CNS_ABI_VERSION 4u /* 2: reservations (cns_resv_soa, `reservation` on running / pending jobs); 3: core ids 128..255
                              (core_w2 / core_w3 planes appended to cns_node_soa, cns_running_soa, cns_resv_soa, cns_placement_soa);
                              4: refusals per group of partitions (cns_node_soa::unsupported appended, CNS_REASON_ENGINE_REFUSED,
                              cns_get_partition_status) and several devices (cns_group_*, cns_comm_*) */
#define CNS_MAX_GRES_CLASSES 8u
#define CNS_MAX_GRES_NAMES 4u
#define CNS_MAX_NODE_TYPES 64u /* distinct res_total records per cycle */
#define CNS_NODE_NONE 0xFFFFFFFFu
#define CNS_RESV_NONE 0xFFFFFFFFu /* job / running job outside every reservation */
#define CNS_TIME_INFINITE_FUTURE INT64_MAX /* absl::InfiniteFuture(), JobScheduler.h:302 */

typedef enum cns_status {
  CNS_OK = 0,
  CNS_ERR_INVALID_ARG = -1,
  CNS_ERR_NO_DEVICE = -2,   /* no usable HIP device: the engine never falls back to CPU */
  CNS_ERR_HIP = -3,         /* a HIP runtime call failed; see cns_last_error */
  CNS_ERR_UNSUPPORTED = -4, /* input outside the engine's documented limits */
  CNS_ERR_STATE = -5,       /* call order violated (e.g. select before set_nodes) */
  CNS_ERR_DEVICE_FAULT = -6 /* kernel reported an internal invariant violation */
} cns_status;

/* Pending reason codes <-> the reference's reason strings
 * (JobScheduler.h:133,138; values set at JobScheduler.cpp:6750-6831, JobScheduler.h:198). */
typedef enum cns_reason {
  CNS_REASON_NONE = 0,               /* ""  : is_scheduled(), starts now            */
  CNS_REASON_PRIORITY = 1,           /* "Priority"                                   */
  CNS_REASON_RESOURCE = 2,           /* "Resource"                                   */
  CNS_REASON_RESOURCE_RESERVED = 3,  /* "Resource Reserved" (JobScheduler.cpp:6799-6806) */
  CNS_REASON_PARTITION_NOT_FOUND = 4,/* "Partition Not Found"                        */
  CNS_REASON_SKIPPED = 5,            /* caller pre-set a reason (e.g. "License")     */
  CNS_REASON_RESERVATION_NOT_FOUND = 6, /* "Reservation Not Found" (JobScheduler.cpp:6756-6758) */
  /* 7: CNS_REASON_PREEMPTED, "Preempted" (include/crane_gpu/preempt.h) */
  CNS_REASON_ENGINE_REFUSED = 8      /* NOT a reason of the reference: the job's partition (its group of partitions connected through
                                        shared nodes) lies outside the engine's limits — nothing was decided for the job, the caller's
                                        CPU SchedulerAlgo takes exactly these jobs (cns_get_partition_status says why) */
} cns_reason;

/* Why a partition is not served (cns_get_partition_status).  The reference bounds none of this (CpuSet is a std::set<uint32_t>, GRES
 * maps are unbounded: PublicHeader.h:555-573,427-494); what the engine cannot hold refuses the GROUP of partitions that touches it and
 * nothing else. */
typedef enum cns_partition_status {
  CNS_PART_SERVED = 0,
  CNS_PART_REFUSED_NODE = 1,   /* it (or a partition it shares a node with) lists a schedulable node flagged cns_node_soa::unsupported */
  CNS_PART_REFUSED_CPU = 2,    /* ... a schedulable node whose cpu_total_raw is outside (0, 2^31-2) */
  CNS_PART_REFUSED_TYPES = 3,  /* its nodes would bring the snapshot's distinct res_total records above CNS_MAX_NODE_TYPES */
  CNS_PART_REFUSED_WIDTH = 4   /* more schedulable (partition, node) slots than the widest tile holds */
} cns_partition_status;

/* Scheduler constants (JobScheduler.h:266-270, CtldPublicDefs.h:82-83). */
typedef struct cns_config {
  uint32_t abi_version;           /* must be CNS_ABI_VERSION                          */
  int32_t device;                 /* HIP device ordinal                               */
  uint64_t scheduled_batch_size;  /* g_config.ScheduledBatchSize; 0 = unlimited       */
  uint32_t max_job_num_per_node;  /* kAlgoMaxJobNumPerNode; 0 -> 1000                 */
  uint32_t kernel_pin;            /* cns_kernel_pin: 0 = the engine chooses per launch (k_wide where every workgroup of the launch is
                                     resident at once, else k_pipe / k_select).  k_wide's workgroups spin on each other: it wants the GPU to
                                     itself — a second process on the device turns it into the bounded-wait + retry path — so a
                                     controller that SHARES its GPU pins CNS_KERNEL_PIPE                            */
  int64_t max_time_window_sec;    /* kAlgoMaxTimeWindow; 0 -> 7*24*3600               */
} cns_config;

typedef enum cns_kernel_pin { CNS_KERNEL_AUTO = 0, CNS_KERNEL_SELECT = 1 /* one workgroup per partition, one wave carries test + commit */,
                               CNS_KERNEL_PIPE = 2 /* one workgroup per partition, decoupled test / commit */ } cns_kernel_pin;

/* (name,type) -> slot-bit layout of the 64-bit GRES mask; fixed per handle cycle. */
typedef struct cns_gres_layout {
  uint32_t num_classes;                       /* <= CNS_MAX_GRES_CLASSES              */
  uint8_t class_name[CNS_MAX_GRES_CLASSES];   /* name-group id < CNS_MAX_GRES_NAMES   */
  uint8_t class_shift[CNS_MAX_GRES_CLASSES];
  uint8_t class_width[CNS_MAX_GRES_CLASSES];  /* classes must not overlap             */
} cns_gres_layout;

/* Per-cycle node snapshot = what NodeSelect's prologue copies out of
 * CranedMetaContainer (JobScheduler.cpp:6563-6617; CranedMeta NodeDefs.h:59-81). */
typedef struct cns_node_soa {
  uint32_t num_nodes;
  uint32_t num_partitions;
  const int64_t* cpu_total_raw;   /* [num_nodes] res_total cpu_count raw             */
  const uint64_t* mem_total;      /* [num_nodes]                                     */
  const uint64_t* core_lo;        /* [num_nodes] res_total core ids 0..63            */
  const uint64_t* core_hi;        /* [num_nodes] core ids 64..127 (may be NULL = 0)  */
  const uint64_t* gres_slots;     /* [num_nodes] may be NULL = 0                     */
  const uint8_t* schedulable;     /* [num_nodes] alive && !drain (JobScheduler.cpp:6595); NULL = all */
  const uint32_t* part_offsets;   /* [num_partitions+1] CSR into part_nodes          */
  const uint32_t* part_nodes;     /* node indices of each partition; a node may be listed by several partitions:
                                     one NodeState per node, one cost per (partition, node), JobScheduler.cpp:6585-6617 */
  cns_gres_layout gres;
  const uint64_t* core_w2;        /* [num_nodes] core ids 128..191 (may be NULL = 0); CpuSet::core_ids is a set of
                                     uint32 ids without a bound, PublicHeader.h:555-573: 256 ids are what the engine carries */
  const uint64_t* core_w3;        /* [num_nodes] core ids 192..255 (may be NULL = 0) */
  const uint8_t* unsupported;     /* [num_nodes] non-zero: the caller could not express this node in the arrays above (a core id >= 256,
                                     more GRES slots / classes than the 64-bit mask holds): the partitions that list it — and those
                                     connected to them through shared nodes — are refused, all others are served.  NULL = none */
} cns_node_soa;

/* Running jobs' per-node allocations (RnJobInScheduler, JobScheduler.h:57-90;
 * folded at JobScheduler.cpp:6513-6514,6681-6709). Order is significant: the
 * initial fp64 cost is accumulated in this order (JobScheduler.h:508-510). */
typedef struct cns_running_soa {
  uint32_t num_jobs;
  uint32_t num_allocs;
  const int64_t* end_sec;         /* [num_jobs] end_time                             */
  const uint32_t* alloc_offsets;  /* [num_jobs+1] CSR into alloc_*                   */
  const uint32_t* alloc_node;
  const int64_t* alloc_cpu_raw;
  const uint64_t* alloc_mem;
  const uint64_t* alloc_core_lo;
  const uint64_t* alloc_core_hi;  /* may be NULL */
  const uint64_t* alloc_gres;     /* may be NULL */
  const uint32_t* reservation;    /* [num_jobs] reservation index the job runs in (JobScheduler.cpp:6692-6707) or CNS_RESV_NONE; NULL = none */
  const uint64_t* alloc_core_w2;  /* core ids 128..191 / 192..255 of the allocation; may be NULL */
  const uint64_t* alloc_core_w3;
} cns_running_soa;

/* Reservations of the cycle = what NodeSelect reads from g_meta_container->GetResvMetaMapPtr()
 * (JobScheduler.cpp:6619-6679; ResvMeta: start_time, end_time, res_total per node).  Semantics kept:
 *   now >= end_sec            : ignored ("expired but not cleaned up", :6631-6634);
 *   start_sec <= now < end    : ACTIVE — its per-node resources count as allocated on the real node until end_sec
 *                               (:6644-6652), and the reservation gets its own scheduler over virtual nodes whose
 *                               res_total is the reserved share and whose time map ends at end_sec (:6657-6668,
 *                               InitTimeAvailResMap(now, end) JobScheduler.h:301-338);
 *   now < start_sec           : FUTURE — a dip [start,end) in the real node's time map and cost (:6669-6677,
 *                               JobScheduler.h:305-308,502-506).
 * Every non-expired reservation marks its nodes for the "Resource Reserved" pending reason (:6635-6642,6799-6806).
 * Canonicalisation: the reference iterates a hash map of reservations (order unspecified, and the initial fp64
 * node cost depends on it); here reservations are applied in ascending index. */
typedef struct cns_resv_soa {
  uint32_t num_resv;
  uint32_t num_allocs;
  const int64_t* start_sec;       /* [num_resv]                                      */
  const int64_t* end_sec;         /* [num_resv]                                      */
  const uint32_t* alloc_offsets;  /* [num_resv+1] CSR: res_total.EachNodeResMap()    */
  const uint32_t* alloc_node;     /* distinct inside one reservation                 */
  const int64_t* alloc_cpu_raw;
  const uint64_t* alloc_mem;
  const uint64_t* alloc_core_lo;
  const uint64_t* alloc_core_hi;  /* may be NULL */
  const uint64_t* alloc_gres;     /* may be NULL */
  const uint64_t* alloc_core_w2;  /* core ids 128..191 / 192..255 of the reserved share; may be NULL */
  const uint64_t* alloc_core_w3;
} cns_resv_soa;

/* Pending jobs in priority order (PdJobInScheduler, JobScheduler.h:92-170).
 * FIFO = ascending job id = input order (BasicPriority, JobScheduler.h:183-201). */
typedef struct cns_job_soa {
  uint64_t num_jobs;
  const uint32_t* partition;        /* [J] partition index; >= num_partitions => "Partition Not Found" */
  const int64_t* time_limit_sec;    /* [J] > 0                                       */
  const int64_t* node_cpu_raw;      /* [J] req_node_res_view cpu (reference asserts 0, JobScheduler.cpp:7047); NULL = 0 */
  const uint64_t* node_mem;         /* [J] req_node_res_view mem                     */
  const int64_t* task_cpu_raw;      /* [J] req_task_res_view cpu                     */
  const uint64_t* task_mem;         /* [J] req_task_res_view mem                     */
  const uint32_t* node_num;         /* [J] >= 1                                      */
  const uint32_t* ntasks;           /* [J] >= node_num                               */
  const uint32_t* ntasks_per_node_min; /* [J] >= 1 (finalised, JobScheduler.cpp:7125-7139) */
  const uint32_t* ntasks_per_node_max; /* [J] >= min                                 */
  const uint8_t* exclusive;         /* [J] NULL = 0                                  */
  const uint8_t* gres_total;        /* [J][CNS_MAX_GRES_NAMES] per-name GresCount.total of req_node_res_view; NULL = none */
  const uint8_t* gres_spec;         /* [J][CNS_MAX_GRES_CLASSES] per-class GresCount.specified; NULL = none */
  const uint64_t* incl_offsets;     /* [J+1] CSR included_nodes; NULL = none         */
  const uint32_t* incl_nodes;
  const uint64_t* excl_offsets;     /* [J+1] CSR excluded_nodes; NULL = none         */
  const uint32_t* excl_nodes;
  const uint8_t* skip;              /* [J] non-zero: reason already set by the caller (JobScheduler.cpp:6744); NULL = 0 */
  const uint32_t* reservation;      /* [J] reservation index the job is submitted to (then `partition` is not looked at,
                                       JobScheduler.cpp:6525-6527,6754-6760) or CNS_RESV_NONE; NULL = none.  An index
                                       >= num_resv, or a reservation that is not active, gives "Reservation Not Found" */
} cns_job_soa;

/* Results, caller-allocated. Job j owns records [place_offsets[j], place_offsets[j+1]),
 * place_offsets = exclusive prefix sum of node_num (the engine fills it).  Records of
 * one job are sorted by ascending node_idx; unused records carry CNS_NODE_NONE.
 * For a job whose reason is RESOURCE with start_sec == 0 nothing was committed. */
typedef struct cns_placement_soa {
  uint64_t place_capacity;   /* >= sum(node_num)                                    */
  int64_t* start_sec;        /* [J] 0 if no start time was found                    */
  uint8_t* reason;           /* [J] cns_reason                                      */
  uint64_t* place_offsets;   /* [J+1]                                               */
  uint32_t* node_idx;        /* [place_capacity]                                    */
  uint32_t* ntasks;          /* craned_id_to_task_num                               */
  int64_t* cpu_raw;          /* allocated_res per node                              */
  uint64_t* mem;
  uint64_t* core_lo;
  uint64_t* core_hi;
  uint64_t* gres;
  uint64_t* core_w2;         /* allocated core ids 128..191 / 192..255; may be NULL unless a node of the snapshot has a  */
  uint64_t* core_w3;         /* core id above 127 (then cns_download fails with CNS_ERR_INVALID_ARG)                      */
} cns_placement_soa;

/* Timing of the last cns_select / cns_run_resident (HIP events on the engine's stream). */
typedef struct cns_timing {
  double h2d_ms;          /* job table pack + upload                                 */
  double init_ms;         /* node-state init kernel(s)                               */
  double select_ms;       /* the persistent selection kernel                         */
  double d2h_ms;          /* placement download                                      */
  uint64_t jobs_ordered;  /* jobs given to the ordered loop (JobScheduler.cpp:6743)  */
  uint64_t algorithmic_bytes; /* sum over ordered jobs of N_p*S_node + S_job + S_out */
} cns_timing;

typedef struct cns_engine cns_handle;

/* ---- several devices ------------------------------------------------------------------------------------------------
 * The reference owns the algorithm through ONE object built at one place and called once per cycle
 * (JobScheduler.cpp:158-159,1441); independent LocalSchedulers per partition are its natural shards (:6723-6732,
 * 6746-6761).  The unit of sharding is the group of partitions connected through shared nodes (one NodeState per
 * craned, :6563,6609-6617); group g runs on device g % N, its slice of the queue keeps its order, and ONE all-gather of
 * the packed result buffers per cycle leaves every device with the merged claim list — RCCL's ncclAllGather (ring over
 * xGMI) whenever every rank has a device of its own; groups never interact, so the merge has no claim to resolve.
 *   cns_group_*            one process drives N devices (the C++ adapter: GpuNodeSelectionAlgo(std::vector<int>));
 *   cns_comm_* + cns_allgather_results   one process per device; the caller ships rank 0's id to the other ranks. */
#define CNS_COMM_ID_BYTES 128u
typedef enum cns_gather_mode {
  CNS_GATHER_RCCL = 1,            /* ncclAllGather, in place, one call per cycle                                    */
  CNS_GATHER_DEVICE_COPIES = 2    /* the group lists a device ordinal twice (RCCL refuses two ranks on one device: how
                                     the path is exercised on a one-GPU box): the same bytes, device-to-device copies */
} cns_gather_mode;
typedef struct cns_results_offsets { /* byte offsets of the packed result buffer (cns_device_results) of the last upload */
  uint64_t num_jobs, num_places;
  uint32_t wide_cores, reserved0;
  uint64_t start_sec, cpu_raw, mem, core_lo, core_hi, gres, node_idx, ntasks, reason, core_w2, core_w3, total_bytes;
} cns_results_offsets;
typedef struct cns_group_info {     /* the last cns_group_select */
  uint32_t num_devices;
  uint32_t gather_mode;             /* cns_gather_mode */
  double shards_ms;                 /* deal + pack + upload + run of every device (host threads, wall clock) */
  double max_select_ms;             /* the slowest device's selection kernel (HIP events) */
  double allgather_ms, download_ms, scatter_ms;
  uint64_t slot_bytes;              /* bytes every rank contributes to the all-gather (padded to the largest shard) */
} cns_group_info;
typedef struct cns_group cns_group;

int cns_abi_version(void);
/* Human readable message of the last error on this handle (or of the last failed cns_create when h==NULL). */
const char* cns_last_error(const cns_handle* h);

int cns_create(const cns_config* cfg, cns_handle** out);
void cns_destroy(cns_handle* h);

/* Per-cycle snapshot. Copies everything; the caller keeps ownership of its buffers. */
int cns_set_nodes(cns_handle* h, const cns_node_soa* nodes);
int cns_set_reservations(cns_handle* h, const cns_resv_soa* resv);  /* after cns_set_nodes, before cns_set_running; NULL: none */
int cns_set_running(cns_handle* h, const cns_running_soa* running); /* NULL or num_jobs==0: none */

/* One scheduling cycle: pack+upload jobs, init node state, select, download placements.
 * Equivalent of SchedulerAlgo::NodeSelect(now, running_jobs, pending_jobs).
 * Synchronous; one caller thread per handle (JobScheduler.cpp:1322).  For queues of 32 768 jobs and more the engine's own pass over
 * the queue (BasicPriority's truncation JobScheduler.h:185-200, the pre-checks JobScheduler.cpp:6744-6761, the split by partition
 * :6516-6530) runs on up to 16 short-lived host threads inside the call while the job arrays are on their way to the device;
 * cns_set_host_threads(h, n) sets their number (1: the calling thread only; 0, the default: CNS_HOST_THREADS from the environment,
 * else up to 16).  The result does not depend on it. */
int cns_set_host_threads(cns_handle* h, uint32_t n);   /* n <= 64; a group: per device, through cns_group_handle */
int cns_select(cns_handle* h, int64_t now_sec, const cns_job_soa* jobs, cns_placement_soa* out);

/* Split form used by the benchmark so that the timed region starts with inputs resident in HBM. */
int cns_upload_jobs(cns_handle* h, const cns_job_soa* jobs);
int cns_run_resident(cns_handle* h, int64_t now_sec);          /* init + select on device, synchronous */
int cns_download(cns_handle* h, cns_placement_soa* out);
/* Device pointer + byte size of the packed placement buffer of the last run (for RCCL allgather). */
int cns_get_partition_status(const cns_handle* h, uint8_t* status /* [num_partitions] cns_partition_status */, uint32_t capacity);
int cns_device_results(cns_handle* h, void** dptr, uint64_t* bytes);
int cns_results_layout(const cns_handle* h, cns_results_offsets* out);   /* where the arrays sit in that buffer */
/* one process per device: a communicator over the ranks' engines, and the all-gather of their packed results */
int cns_comm_unique_id(uint8_t id[CNS_COMM_ID_BYTES]);                    /* rank 0; ship the bytes to every rank */
int cns_comm_init_rank(cns_handle* h, uint32_t nranks, uint32_t rank, const uint8_t id[CNS_COMM_ID_BYTES]);   /* collective */
int cns_comm_destroy(cns_handle* h);
int cns_allgather_results(cns_handle* h, uint64_t slot_bytes, void** gathered_dptr);   /* collective, after a run: rank r's buffer
                                                                                          (padded to slot_bytes, a multiple of 16) at [r * slot_bytes) */
int cns_download_gathered(cns_handle* h, void* dst, uint64_t bytes);
int cns_gather_timing(const cns_handle* h, double* ms, uint64_t* bytes);
/* one process, N devices */
int cns_group_create(const cns_config* cfg, const int32_t* devices, uint32_t num_devices, cns_group** out);   /* cfg->device is ignored */
void cns_group_destroy(cns_group* g);
const char* cns_group_last_error(const cns_group* g);
uint32_t cns_group_size(const cns_group* g);
cns_handle* cns_group_handle(cns_group* g, uint32_t device_index);          /* e.g. for cns_host_alloc, cns_debug_* of one device */
int cns_group_set_nodes(cns_group* g, const cns_node_soa* nodes);          /* deals the groups of partitions: group i -> device i % N */
int cns_group_set_reservations(cns_group* g, const cns_resv_soa* resv);    /* every device; the jobs of reservation v run on (active) device v % N */
int cns_group_set_running(cns_group* g, const cns_running_soa* running);   /* every device (allocations on nodes it does not schedule are dropped there) */
int cns_group_select(cns_group* g, int64_t now_sec, const cns_job_soa* jobs, cns_placement_soa* out);   /* = cns_select, merged in queue order: cns_config::
                                       scheduled_batch_size cuts the ONE ordered queue (JobScheduler.h:185-200), not every device's share of it */
/* = cns_get_partition_status over the caller's partitions.  A device whose WHOLE share of the partitions is outside the engine's limits serves
 * nothing (its jobs: CNS_REASON_ENGINE_REFUSED) while the other devices run; cns_group_set_nodes fails only when no device can serve anything. */
int cns_group_get_partition_status(const cns_group* g, uint8_t* status /* [num_partitions] cns_partition_status */, uint32_t capacity);
int cns_group_get_info(const cns_group* g, cns_group_info* out);
uint32_t cns_group_device_of_partition(const cns_group* g, uint32_t partition);

/* Page-locked host memory for the caller's job arrays and result arrays (optional).  Every host buffer stays the caller's
 * (ownership); buffers from here are copied by the DMA engines directly — pageable memory goes through the
 * runtime's staging buffers at a third of the rate, and a freshly allocated result array pays a page fault per 4 KB on its
 * first download.  Keep them across cycles (the adapter packs the job table into them; JobScheduler.cpp:1439-1447 is the
 * bracket they shorten).  cns_host_free, or cns_destroy, releases them. */
int cns_host_alloc(cns_handle* h, uint64_t bytes, void** out);
int cns_host_free(cns_handle* h, void* p);

int cns_get_timing(const cns_handle* h, cns_timing* t);

/* Parity / debugging: final per-node cost (fp64 bit patterns) and time-availability map. */
int cns_debug_get_costs(cns_handle* h, double* cost_by_part_slot /* [len(part_nodes)] */);
int cns_debug_get_timeline(cns_handle* h, uint32_t node, uint32_t capacity, uint32_t* len,
                           int64_t* t, int64_t* cpu_raw, uint64_t* mem, uint64_t* core_lo,
                           uint64_t* core_hi, uint64_t* gres);
/* ... and the core ids 128..255 of the same entries (0 for nodes without any). */
int cns_debug_get_timeline_cores(cns_handle* h, uint32_t node, uint32_t capacity, uint64_t* core_w2, uint64_t* core_w3);

/* Name of the selection kernel the last cns_run_resident launched ("k_pipe<16>", "k_select<19>", ...; "" before a run). */
const char* cns_debug_last_kernel(const cns_handle* h);

/* Cycle counters of the last run (32 per partition); zeros unless the library was built with -DCNS_PROF. */
int cns_debug_get_prof(cns_handle* h, uint64_t* out, uint32_t capacity);
uint32_t cns_debug_engine_partitions(const cns_handle* h);   /* schedulers of the snapshot as the engine runs them: groups of partitions that share nodes + one per reservation (the rows of cns_debug_get_prof) */

#ifdef __cplusplus
}
#endif
#endif /* CRANE_GPU_NODE_SELECT_H_ */
