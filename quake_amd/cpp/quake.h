// quake.h -- umbrella header for sources written against the reference's headers (quake_index.h, query_coordinator.h,
// partition_manager.h, list_scanning.h, common.h): the mirror's classes in the global namespace, and the two
// faiss::MetricType names the reference's call sites use.  Include this instead of the reference's headers.
#pragma once
#include "quake_index.h"

#include <chrono>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

using namespace quake_amd;  // the reference declares its classes at global scope
// the unqualified std names the reference's common.h brings in (common.h:49-61)
using std::make_shared;
using std::shared_ptr;
using std::size_t;
using std::string;
using std::tuple;
using std::unordered_map;
using std::vector;
using std::chrono::duration_cast;
using std::chrono::high_resolution_clock;
using std::chrono::microseconds;
using std::chrono::milliseconds;
using std::chrono::nanoseconds;
namespace faiss {
using MetricType = quake_amd::MetricType;
constexpr MetricType METRIC_INNER_PRODUCT = quake_amd::METRIC_INNER_PRODUCT;
constexpr MetricType METRIC_L2 = quake_amd::METRIC_L2;
using idx_t = int64_t;
}  // namespace faiss
