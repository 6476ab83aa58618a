// quake.h -- umbrella header for sources written against the reference's headers (quake_index.h, query_coordinator.h,
// partition_manager.h, list_scanning.h, common.h): the mirror's classes in the global namespace, and the two
// faiss::MetricType names the reference's call sites use.  Include this instead of the reference's headers.
#pragma once
#include "quake_index.h"

using namespace quake_amd;  // the reference declares its classes at global scope
namespace faiss {
using MetricType = quake_amd::MetricType;
constexpr MetricType METRIC_INNER_PRODUCT = quake_amd::METRIC_INNER_PRODUCT;
constexpr MetricType METRIC_L2 = quake_amd::METRIC_L2;
using idx_t = int64_t;
}  // namespace faiss
