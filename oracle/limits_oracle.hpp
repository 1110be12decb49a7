// CPU ORACLE (TEST INFRASTRUCTURE ONLY — never linked into or called by the product path).
//
// Restatement of the run-limit admission that JobScheduler::ScheduleThread_ applies to the jobs NodeSelect
// started.  Function by function after the reference (CraneSched tree):
//   commit loop                                 src/CraneCtld/JobScheduler.cpp:1492-1573
//   CheckAndMallocMetaResource                  src/CraneCtld/Accounting/AccountMetaContainer.cpp:180-224
//   CheckRunLimits_                             :891-1028
//   CheckQosRunLimitsForEntity_                 :508-540
//   CheckPartitionRunLimitsForEntity_           :542-670
//   CheckTres_ / IsUnlimitedTres_ / CheckGres_  :345-365,1030-1050
//   DoMallocResource_                           :1067-1124
//   ResourceView += / GresCount +=              src/Utilities/PublicHeader/PublicHeader.cpp:23-29,448-456
//
// The reference's maps (std::unordered_map keyed by strings) are kept as MAPS here — entries appear and are
// looked up exactly where the reference does — but keyed by the dense indices of include/crane_gpu/run_limits.h
// and ordered (std::map), which fixes the iteration order CheckGres_ depends on (ascending name, ascending
// type class).  PINNED (round 4): the reference holds no test for this path (test/ has no AccountMetaContainer case),
// but its own code runs here: oracle/_ref compiles AccountMetaContainer.h:30-295 and AccountMetaContainer.cpp:39-45,
// 180-224,345-365,508-687,891-1124 (sliced at build time by oracle/ref_build/extract.py) and
// tests/test_ref_pin_limits_steps.py holds this restatement to it — every reason string, the admitted count and every
// usage record after the pass, on the hand-derived scenarios of tests/test_run_limits.py, 40 random account trees and
// BASELINE config 4's account / QoS tables.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "../include/crane_gpu/run_limits.h"

namespace lim_oracle {

struct GresCount {  // PublicHeader.h:505-507
  uint64_t total = 0;
  std::map<uint32_t, uint64_t> specified;  // class index -> count
};
using GresMap = std::map<uint32_t, GresCount>;  // name index -> GresCount

struct ResourceView {  // cpu_t raw, memory bytes, GresMap (mem_sw is never tested on this path)
  int64_t cpu = 0;
  uint64_t mem = 0;
  GresMap gres;
  ResourceView& operator+=(const ResourceView& r) {  // PublicHeader.cpp:448-456, GresCount += :23-29
    cpu += r.cpu;
    mem += r.mem;
    for (const auto& [name, gc] : r.gres) {
      GresCount& d = gres[name];
      d.total += gc.total;
      for (const auto& [type, cnt] : gc.specified) d.specified[type] += cnt;
    }
    return *this;
  }
};

struct MetaResource {  // AccountMetaContainer.h:30-35
  ResourceView resource;
  uint32_t jobs_count = 0;
  int64_t wall_time = 0;
  MetaResource& operator+=(const MetaResource& r) {
    resource += r.resource;
    jobs_count += r.jobs_count;
    wall_time += r.wall_time;
    return *this;
  }
};

struct Qos {  // AccountDefs.h:33-45
  uint32_t max_jobs_per_user, max_jobs_per_account, max_jobs;
  int64_t max_cpus_per_user, max_wall;
  ResourceView max_tres, max_tres_per_user, max_tres_per_account;
};
struct PartitionResourceLimit {  // AccountDefs.h:163-175
  ResourceView max_tres;
  uint32_t max_jobs;
  int64_t max_wall;
};
struct MetaResourceStat {  // one user or one account (AccountMetaContainer.h:62-80)
  std::map<uint32_t, MetaResource> qos_to_resource_map;
  std::map<uint32_t, std::map<uint32_t, MetaResource>> account_to_partition_to_resource_map;  // users (key: user_acct)
  std::map<uint32_t, MetaResource> partition_to_resource_map;                                  // accounts
};

inline const char* reason_string(int code) {
  static const char* s[] = {"", "QosEntryNotFound", "QosCpuResourceLimit", "QosJobsResourceLimit", "QosWallTimeLimit",
                            "QosCpuResourceLimit", "QosMemResourceLimit", "QosGresResourceLimit", "PartitionEntryNotFound",
                            "UserPartitionJobsLimit", "UserPartitionWallTimeLimit", "AccPartitionJobsLimit",
                            "AccPartitionWallTimeLimit", "PartitionCpuResourceLimit", "PartitionMemResourceLimit",
                            "PartitionGresResourceLimit"};
  return code >= 0 && code < 16 ? s[code] : "?";
}

class Limits {
 public:
  // ---- construction from the C-ABI tables -------------------------------------------------------------
  Limits(const cns_limit_tables& t, const cns_gres_layout& gl) : gl_(gl), Q_(t.num_qos), Pn_(t.num_partitions) {
    for (uint32_t q = 0; q < t.num_qos; ++q) {
      const cns_qos_limits& s = t.qos[q];
      qos_.push_back(Qos{s.max_jobs_per_user, s.max_jobs_per_account, s.max_jobs, s.max_cpus_per_user_raw,
                         s.max_wall_sec, view_of(s.max_tres), view_of(s.max_tres_per_user),
                         view_of(s.max_tres_per_account)});
    }
    for (uint32_t i = 0; i < t.num_part_limits; ++i)
      part_limits_.push_back(PartitionResourceLimit{view_of(t.part_limits[i].max_tres), t.part_limits[i].max_jobs,
                                                    t.part_limits[i].max_wall_sec});
    parent_.assign(t.acct_parent, t.acct_parent + t.num_accounts);
    if (t.user_part_limit) user_part_limit_.assign(t.user_part_limit, t.user_part_limit + (size_t)t.num_user_accts * Pn_);
    if (t.acct_part_limit) acct_part_limit_.assign(t.acct_part_limit, t.acct_part_limit + (size_t)t.num_accounts * Pn_);
    users_.resize(t.num_users);
    accounts_.resize(t.num_accounts);
    user_of_ua_.assign(t.num_user_accts, CNS_LIM_NONE);
    auto ex = [](const uint8_t* e, size_t i) { return !e || e[i]; };
    auto us = [&](const cns_usage* u, size_t i) { return u ? meta_of(u[i]) : MetaResource{}; };
    for (uint32_t u = 0; u < t.num_users; ++u)
      for (uint32_t q = 0; q < Q_; ++q)
        if (ex(t.user_qos_exists, (size_t)u * Q_ + q)) users_[u].qos_to_resource_map[q] = us(t.user_qos, (size_t)u * Q_ + q);
    // user_part entries are attached to their user lazily (the pair -> user relation arrives with the jobs)
    ua_part_init_.resize((size_t)t.num_user_accts * Pn_);
    ua_part_exists_.resize((size_t)t.num_user_accts * Pn_);
    for (size_t i = 0; i < ua_part_init_.size(); ++i) {
      ua_part_exists_[i] = ex(t.user_part_exists, i);
      ua_part_init_[i] = us(t.user_part, i);
    }
    for (uint32_t a = 0; a < t.num_accounts; ++a) {
      for (uint32_t q = 0; q < Q_; ++q)
        if (ex(t.acct_qos_exists, (size_t)a * Q_ + q)) accounts_[a].qos_to_resource_map[q] = us(t.acct_qos, (size_t)a * Q_ + q);
      for (uint32_t p = 0; p < Pn_; ++p)
        if (ex(t.acct_part_exists, (size_t)a * Pn_ + p))
          accounts_[a].partition_to_resource_map[p] = us(t.acct_part, (size_t)a * Pn_ + p);
    }
    for (uint32_t q = 0; q < Q_; ++q) qos_meta_.push_back(us(t.qos_usage, q));
  }

  // ---- the admission of one job: CheckAndMallocMetaResource (:180-224) ----------------------------------
  // alloc = job.allocated_res.View(); returns a cns_limit_reason
  int check_and_malloc(uint32_t user, uint32_t ua, uint32_t account, uint32_t qos, uint32_t part, int64_t time_limit,
                       const ResourceView& alloc) {
    attach_user_acct(user, ua);
    std::vector<uint32_t> chain;  // job.account_chain: the job's account first, then its ancestors
    for (uint32_t a = account; a != CNS_LIM_NONE; a = parent_[a]) chain.push_back(a);
    int r = check_run_limits(user, ua, chain, qos, part, time_limit, alloc);
    if (r) return r;
    MetaResource meta;  // :212-215  {allocated view, jobs_count 1, wall_time = time_limit}
    meta.resource = alloc;
    meta.jobs_count = 1;
    meta.wall_time = time_limit;
    do_malloc(user, ua, chain, qos, part, meta);
    return 0;
  }

  // ---- state read-back in the C-ABI shape --------------------------------------------------------------
  void export_usage(uint32_t num_users, uint32_t num_uas, cns_usage* uq, uint8_t* uqe, cns_usage* up, uint8_t* upe,
                    cns_usage* aq, uint8_t* aqe, cns_usage* ap, uint8_t* ape, cns_usage* qu) const {
    for (uint32_t u = 0; u < num_users; ++u)
      for (uint32_t q = 0; q < Q_; ++q) {
        auto it = users_[u].qos_to_resource_map.find(q);
        put(uq, uqe, (size_t)u * Q_ + q, it == users_[u].qos_to_resource_map.end() ? nullptr : &it->second);
      }
    for (uint32_t x = 0; x < num_uas; ++x)
      for (uint32_t p = 0; p < Pn_; ++p) {
        const MetaResource* m = nullptr;
        MetaResource tmp;
        if (user_of_ua_[x] != CNS_LIM_NONE) {
          const auto& mm = users_[user_of_ua_[x]].account_to_partition_to_resource_map;
          auto a = mm.find(x);
          if (a != mm.end()) { auto b = a->second.find(p); if (b != a->second.end()) m = &b->second; }
        } else if (ua_part_exists_[(size_t)x * Pn_ + p]) {
          tmp = ua_part_init_[(size_t)x * Pn_ + p];
          m = &tmp;
        }
        put(up, upe, (size_t)x * Pn_ + p, m);
      }
    for (uint32_t a = 0; a < accounts_.size(); ++a) {
      for (uint32_t q = 0; q < Q_; ++q) {
        auto it = accounts_[a].qos_to_resource_map.find(q);
        put(aq, aqe, (size_t)a * Q_ + q, it == accounts_[a].qos_to_resource_map.end() ? nullptr : &it->second);
      }
      for (uint32_t p = 0; p < Pn_; ++p) {
        auto it = accounts_[a].partition_to_resource_map.find(p);
        put(ap, ape, (size_t)a * Pn_ + p, it == accounts_[a].partition_to_resource_map.end() ? nullptr : &it->second);
      }
    }
    for (uint32_t q = 0; q < Q_; ++q) put(qu, nullptr, q, &qos_meta_[q]);
  }

  ResourceView view_of_counts(int64_t cpu, uint64_t mem, const uint64_t* class_count) const {
    // ResourceView += DedicatedResourceInNode (PublicHeader.cpp:417-427): per slot type, total and specified grow together
    ResourceView v;
    v.cpu = cpu;
    v.mem = mem;
    for (uint32_t g = 0; g < gl_.num_classes; ++g)
      if (class_count[g]) {
        GresCount& gc = v.gres[gl_.class_name[g]];
        gc.total += class_count[g];
        gc.specified[g] += class_count[g];
      }
    return v;
  }

 private:
  ResourceView view_of(const cns_tres& t) const {
    ResourceView v;
    v.cpu = t.cpu_raw;
    v.mem = t.mem;
    for (uint32_t n = 0; n < CNS_MAX_GRES_NAMES; ++n)
      if (t.name_mask >> n & 1) v.gres[n].total = t.name_total[n];
    for (uint32_t g = 0; g < gl_.num_classes; ++g)
      if ((t.class_mask >> g & 1) && (t.name_mask >> gl_.class_name[g] & 1)) v.gres[gl_.class_name[g]].specified[g] = t.class_count[g];
    return v;
  }
  MetaResource meta_of(const cns_usage& u) const {
    MetaResource m;
    m.resource.cpu = u.cpu_raw;
    m.resource.mem = u.mem;
    for (uint32_t n = 0; n < CNS_MAX_GRES_NAMES; ++n)
      if (u.name_total[n]) m.resource.gres[n].total = u.name_total[n];
    for (uint32_t g = 0; g < gl_.num_classes; ++g)
      if (u.class_count[g]) m.resource.gres[gl_.class_name[g]].specified[g] = u.class_count[g];
    m.jobs_count = u.jobs_count;
    m.wall_time = u.wall_sec;
    return m;
  }
  void put(cns_usage* out, uint8_t* ex, size_t i, const MetaResource* m) const {
    if (ex) ex[i] = m ? 1 : 0;
    if (!out) return;
    cns_usage u{};
    if (m) {
      u.cpu_raw = m->resource.cpu;
      u.mem = m->resource.mem;
      u.wall_sec = m->wall_time;
      u.jobs_count = m->jobs_count;
      for (const auto& [name, gc] : m->resource.gres) {
        u.name_total[name] = gc.total;
        for (const auto& [type, cnt] : gc.specified) u.class_count[type] = cnt;
      }
    }
    out[i] = u;
  }
  void attach_user_acct(uint32_t user, uint32_t ua) {
    if (user_of_ua_[ua] != CNS_LIM_NONE) return;
    user_of_ua_[ua] = user;
    for (uint32_t p = 0; p < Pn_; ++p)
      if (ua_part_exists_[(size_t)ua * Pn_ + p])
        users_[user].account_to_partition_to_resource_map[ua][p] = ua_part_init_[(size_t)ua * Pn_ + p];
  }

  // CheckGres_ (:1030-1050) — note the two `return true`
  static bool check_gres(const GresMap& req, const GresMap& total) {
    for (const auto& [name, lhs] : req) {
      auto rhs_it = total.find(name);
      if (rhs_it == total.end()) return true;
      const GresCount& rhs = rhs_it->second;
      if (lhs.total > rhs.total) return false;
      for (const auto& [type, lhs_cnt] : lhs.specified) {
        auto t = rhs.specified.find(type);
        if (t == rhs.specified.end()) return true;
        if (lhs_cnt > t->second) return false;
      }
    }
    return true;
  }
  // CheckTres_ (:345-360); prefix 0 = "Qos" (the default, AccountMetaContainer.h:179-181), 8 = "Partition" (reason codes 5..7 / 13..15)
  static int check_tres(const ResourceView& req, const ResourceView& total, int prefix) {
    if (req.cpu > total.cpu) return CNS_LIM_CPU + prefix;
    if (req.mem > total.mem) return CNS_LIM_MEM + prefix;
    if (!check_gres(req.gres, total.gres)) return CNS_LIM_GRES + prefix;
    return 0;
  }
  static bool is_unlimited_tres(const ResourceView& r) {  // :362-365
    return r.cpu == CNS_LIM_UNLIMITED_CPU_RAW && r.mem == CNS_LIM_MAX_JOB_MEMORY && r.gres.empty();
  }

  // CheckQosRunLimitsForEntity_ (:508-540)
  int check_qos_entity(const MetaResourceStat& stat, uint32_t qos_id, const Qos& qos, bool is_user,
                       const ResourceView& alloc, int64_t time_limit) const {
    auto it = stat.qos_to_resource_map.find(qos_id);
    if (it == stat.qos_to_resource_map.end()) return CNS_LIM_QOS_ENTRY_NOT_FOUND;
    const MetaResource& val = it->second;
    ResourceView use = alloc;
    use += val.resource;
    if (is_user) {
      if (use.cpu > qos.max_cpus_per_user) return CNS_LIM_QOS_CPU;
      if ((uint64_t)val.jobs_count + 1 > qos.max_jobs_per_user) return CNS_LIM_QOS_JOBS;
      if (qos.max_wall > 0 && val.wall_time + time_limit > qos.max_wall) return CNS_LIM_QOS_WALL;
      return check_tres(use, qos.max_tres_per_user, 0);
    }
    if ((uint64_t)val.jobs_count + 1 > qos.max_jobs_per_account) return CNS_LIM_QOS_JOBS;
    if (qos.max_wall > 0 && val.wall_time + time_limit > qos.max_wall) return CNS_LIM_QOS_WALL;
    return check_tres(use, qos.max_tres_per_account, 0);
  }

  // CheckPartitionRunLimitsForEntity_ (:542-670)
  int check_part_entity(const MetaResourceStat& stat, uint32_t ua, uint32_t part, const PartitionResourceLimit* pl,
                        const ResourceView& alloc, int64_t time_limit, const Qos& qos, bool is_user) const {
    if (!pl) return 0;
    const MetaResource* val = nullptr;
    if (is_user) {
      auto a = stat.account_to_partition_to_resource_map.find(ua);
      if (a == stat.account_to_partition_to_resource_map.end()) return CNS_LIM_PARTITION_ENTRY_NOT_FOUND;
      auto p = a->second.find(part);
      if (p == a->second.end()) return CNS_LIM_PARTITION_ENTRY_NOT_FOUND;
      val = &p->second;
    } else {
      auto p = stat.partition_to_resource_map.find(part);
      if (p == stat.partition_to_resource_map.end()) return CNS_LIM_PARTITION_ENTRY_NOT_FOUND;
      val = &p->second;
    }
    const uint32_t qos_jobs = is_user ? qos.max_jobs_per_user : qos.max_jobs_per_account;
    if (qos_jobs == CNS_LIM_UNLIMITED_JOBS)  // only when the QoS does not already cap running jobs
      if ((uint64_t)val->jobs_count + 1 > pl->max_jobs)
        return is_user ? CNS_LIM_USER_PARTITION_JOBS : CNS_LIM_ACC_PARTITION_JOBS;
    if (qos.max_wall == 0 && pl->max_wall > 0)
      if (val->wall_time + time_limit > pl->max_wall)
        return is_user ? CNS_LIM_USER_PARTITION_WALL : CNS_LIM_ACC_PARTITION_WALL;
    if (is_unlimited_tres(is_user ? qos.max_tres_per_user : qos.max_tres_per_account)) {
      ResourceView use = alloc;
      use += val->resource;
      if (int r = check_tres(use, pl->max_tres, 8)) return r;
    }
    return 0;
  }

  // CheckRunLimits_ (:891-1028): user entity, account chain, global QoS
  int check_run_limits(uint32_t user, uint32_t ua, const std::vector<uint32_t>& chain, uint32_t qos_id, uint32_t part,
                       int64_t time_limit, const ResourceView& alloc) const {
    const Qos& qos = qos_[qos_id];
    {
      const PartitionResourceLimit* pl = nullptr;
      if (!user_part_limit_.empty() && user_part_limit_[(size_t)ua * Pn_ + part] != CNS_LIM_NONE)
        pl = &part_limits_[user_part_limit_[(size_t)ua * Pn_ + part]];
      if (int r = check_qos_entity(users_[user], qos_id, qos, true, alloc, time_limit)) return r;  // CheckEntityRunLimits_ :672-688
      if (int r = check_part_entity(users_[user], ua, part, pl, alloc, time_limit, qos, true)) return r;
    }
    for (uint32_t a : chain) {
      const PartitionResourceLimit* pl = nullptr;
      if (!acct_part_limit_.empty() && acct_part_limit_[(size_t)a * Pn_ + part] != CNS_LIM_NONE)
        pl = &part_limits_[acct_part_limit_[(size_t)a * Pn_ + part]];
      if (int r = check_qos_entity(accounts_[a], qos_id, qos, false, alloc, time_limit)) return r;
      if (int r = check_part_entity(accounts_[a], ua, part, pl, alloc, time_limit, qos, false)) return r;
    }
    const MetaResource& val = qos_meta_[qos_id];  // :985-1025
    ResourceView use = alloc;
    use += val.resource;
    if ((uint64_t)val.jobs_count + 1 > qos.max_jobs) return CNS_LIM_QOS_JOBS;
    if (qos.max_wall > 0 && val.wall_time + time_limit > qos.max_wall) return CNS_LIM_QOS_WALL;
    return check_tres(use, qos.max_tres, 0);
  }

  // DoMallocResource_ (:1067-1124): every map entry is created when missing
  void do_malloc(uint32_t user, uint32_t ua, const std::vector<uint32_t>& chain, uint32_t qos, uint32_t part,
                 const MetaResource& m) {
    users_[user].qos_to_resource_map[qos] += m;
    users_[user].account_to_partition_to_resource_map[ua][part] += m;
    for (uint32_t a : chain) {
      accounts_[a].qos_to_resource_map[qos] += m;
      accounts_[a].partition_to_resource_map[part] += m;
    }
    qos_meta_[qos] += m;
  }

  cns_gres_layout gl_;
  uint32_t Q_, Pn_;
  std::vector<Qos> qos_;
  std::vector<PartitionResourceLimit> part_limits_;
  std::vector<uint32_t> parent_, user_part_limit_, acct_part_limit_, user_of_ua_;
  std::vector<MetaResourceStat> users_, accounts_;
  std::vector<MetaResource> qos_meta_, ua_part_init_;
  std::vector<uint8_t> ua_part_exists_;
};

}  // namespace lim_oracle
