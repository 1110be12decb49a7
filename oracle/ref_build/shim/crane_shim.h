// ORACLE / TEST INFRASTRUCTURE ONLY — the environment the sliced REFERENCE sources expect
// (oracle/ref_build/extract.py), reduced to what the slices touch.  Everything here is a
// stand-in written for this build; none of it is reference code.
//
// Part 1 (before the PublicHeader slice): ids, cpu_t, forward declarations of the protobuf
//         message types that only appear in declarations, the assertion / log macros.
// Part 2 (crane_shim_ctld.h, after the PublicHeader slice and before the JobScheduler slice):
//         JobInCtld as PdJobInScheduler / RnJobInScheduler read it, g_config, and the
//         singletons NodeSelect calls (meta container, account manager, license manager,
//         job scheduler).
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <functional>
#include <iterator>
#include <list>
#include <map>
#include <memory>
#include <optional>
#include <queue>
#include <ranges>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <variant>
#include <vector>

namespace crane_ref {
// A CRANE_ASSERT / ABSL_ASSERT of the reference failed.  (Release builds of the reference
// compile them away, Logger.h:139-150; debug builds terminate.  The harness reports it.)
struct RefAssertion : std::runtime_error {
  using std::runtime_error::runtime_error;
};
inline bool g_asserts_enabled = true;
[[noreturn]] inline void assertion_failed(const char* what) { throw RefAssertion(what); }
}  // namespace crane_ref

#define CRANE_REF_CHECK(cond, text)                                                         \
  do {                                                                                      \
    if (::crane_ref::g_asserts_enabled && !(cond)) ::crane_ref::assertion_failed(text);     \
  } while (false)

#include "absl_shim.h"
#include "expected_shim.h"
#include "fpm/fixed.hpp"

// crane/Logger.h:62-150
#define CRANE_TRACE(...) (void)0
#define CRANE_DEBUG(...) (void)0
#define CRANE_INFO(...) (void)0
#define CRANE_WARN(...) (void)0
#define CRANE_ERROR(...) (void)0
#define CRANE_CRITICAL(...) (void)0
#define CRANE_ASSERT(cond) CRANE_REF_CHECK(cond, "CRANE_ASSERT(" #cond ")")
#define CRANE_ASSERT_MSG(cond, message) CRANE_REF_CHECK(cond, "CRANE_ASSERT_MSG(" #cond ")")

// crane/PublicHeader.h:33-44
using job_id_t = uint32_t;
using task_id_t = uint32_t;
using step_id_t = uint32_t;
using PartitionId = std::string;
using CranedId = std::string;
using ResvId = std::string;
using cpu_t = fpm::fixed<int64_t, __int128, 8>;
using LicenseId = std::string;
inline const char* const kResourceTypeGpu = "gpu";   // crane/PublicHeader.h:137 (ResourceView::GpuCount, not on this path)

// protobuf message types: only named in declarations of conversion members that the slices
// never define or call.
namespace crane::grpc {
class DeviceTypeSlotsMap;
class DedicatedResourceInNode;
class GresCount;
class GresMap;
class ResourceInNodeV3;
class ResourceV3;
class ResourceView;
enum JobStatus { Pending, Running, Completed, Failed, ExceedTimeLimit, Cancelled, OutOfMemory, Deadline, Configuring };
enum class PreemptType { PREEMPT_NONE = 0, PREEMPT_QOS = 1 };
struct JobToCtld {
  struct License {   // message JobToCtld.License { string key = 1; uint32 count = 2; } — what LicenseManager.cpp:183-210 reads of it
    std::string key_;
    uint32_t count_ = 0;
    const std::string& key() const { return key_; }
    uint32_t count() const { return count_; }
  };
  int licenses_count() const { return 0; }
  bool is_licenses_or() const { return false; }
};
}  // namespace crane::grpc

namespace google::protobuf {
template <class T>
class RepeatedPtrField {   // element order = request order; the slices iterate it, test empty() and read size()
 public:
  RepeatedPtrField() = default;
  explicit RepeatedPtrField(int) {}
  int size() const { return (int)v_.size(); }
  bool empty() const { return v_.empty(); }
  typename std::vector<T>::const_iterator begin() const { return v_.begin(); }
  typename std::vector<T>::const_iterator end() const { return v_.end(); }
  T* Add() { v_.emplace_back(); return &v_.back(); }
 private:
  std::vector<T> v_;
};
}  // namespace google::protobuf
