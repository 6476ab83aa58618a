// bindings.cpp -- pybind11 module quake_amd._bindings: the classes, attributes and method names of the reference's
// `quake._bindings` (src/cpp/bindings/wrap.cpp:48-368), bound to the C++ host mirror in quake_index.h.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>
#include <torch/extension.h>

#include <sstream>

#include "quake_index.h"

namespace py = pybind11;
using namespace quake_amd;

PYBIND11_MODULE(_bindings, m) {
    m.doc() = "quake_amd: MI355X-native Quake search path (same surface as quake._bindings)";

    py::class_<QuakeIndex, std::shared_ptr<QuakeIndex>>(m, "QuakeIndex")
        .def(py::init<int>(), py::arg("current_level") = 0)
        .def("build", &QuakeIndex::build, "Build the index from vectors [n, d] and ids [n].")
        .def("search", &QuakeIndex::search, "Search the index for nearest neighbors.")
        .def("get", &QuakeIndex::get)
        .def("get_ids", &QuakeIndex::get_ids)
        .def("add", &QuakeIndex::add)
        .def("remove", &QuakeIndex::remove)
        .def("modify", &QuakeIndex::modify)
        .def("maintenance", &QuakeIndex::maintenance)
        .def("initialize_maintenance_policy", &QuakeIndex::initialize_maintenance_policy)
        .def("refine_partitions", &QuakeIndex::refine_partitions, py::arg("partition_ids"), py::arg("iterations") = 0)
        .def("save", &QuakeIndex::save)
        .def("load", &QuakeIndex::load, py::arg("path"), py::arg("n_workers") = 0)
        .def("ntotal", &QuakeIndex::ntotal)
        .def("nlist", &QuakeIndex::nlist)
        .def("d", &QuakeIndex::d)
        .def_readonly("parent", &QuakeIndex::parent_)
        .def_readonly("current_level", &QuakeIndex::current_level_)
        .def("__repr__", [](const QuakeIndex &q) {
            std::ostringstream oss;
            oss << "{\"current_level\": " << q.current_level_ << ", }";
            return oss.str();
        });

    py::class_<IndexBuildParams, std::shared_ptr<IndexBuildParams>>(m, "IndexBuildParams")
        .def(py::init<>())
        .def_readwrite("nlist", &IndexBuildParams::nlist)
        .def_readwrite("niter", &IndexBuildParams::niter)
        .def_readwrite("metric", &IndexBuildParams::metric)
        .def_readwrite("num_workers", &IndexBuildParams::num_workers)
        .def("__repr__", [](const IndexBuildParams &p) {
            std::ostringstream oss;
            oss << "{\"nlist\": " << p.nlist << ", \"niter\": " << p.niter << ", \"metric\": \"" << p.metric
                << "\", \"num_workers\": " << p.num_workers << "}";
            return oss.str();
        });

    py::class_<SearchParams, std::shared_ptr<SearchParams>>(m, "SearchParams")
        .def(py::init<>())
        .def_readwrite("k", &SearchParams::k)
        .def_readwrite("nprobe", &SearchParams::nprobe)
        .def_readwrite("recall_target", &SearchParams::recall_target)
        .def_readwrite("num_threads", &SearchParams::num_threads)
        .def_readwrite("batched_scan", &SearchParams::batched_scan)
        .def_readwrite("use_precomputed", &SearchParams::use_precomputed)
        .def_readwrite("initial_search_fraction", &SearchParams::initial_search_fraction)
        .def_readwrite("recompute_threshold", &SearchParams::recompute_threshold)
        .def_readwrite("aps_flush_period_us", &SearchParams::aps_flush_period_us);

    py::class_<MaintenancePolicyParams, std::shared_ptr<MaintenancePolicyParams>>(m, "MaintenancePolicyParams")
        .def(py::init<>())
        .def_readwrite("maintenance_policy", &MaintenancePolicyParams::maintenance_policy)
        .def_readwrite("window_size", &MaintenancePolicyParams::window_size)
        .def_readwrite("refinement_radius", &MaintenancePolicyParams::refinement_radius)
        .def_readwrite("refinement_iterations", &MaintenancePolicyParams::refinement_iterations)
        .def_readwrite("min_partition_size", &MaintenancePolicyParams::min_partition_size)
        .def_readwrite("alpha", &MaintenancePolicyParams::alpha)
        .def_readwrite("enable_split_rejection", &MaintenancePolicyParams::enable_split_rejection)
        .def_readwrite("enable_delete_rejection", &MaintenancePolicyParams::enable_delete_rejection)
        .def_readwrite("delete_threshold_ns", &MaintenancePolicyParams::delete_threshold_ns)
        .def_readwrite("split_threshold_ns", &MaintenancePolicyParams::split_threshold_ns);

    py::class_<MaintenanceTimingInfo, std::shared_ptr<MaintenanceTimingInfo>>(m, "MaintenanceTimingInfo")
        .def_readonly("total_time_us", &MaintenanceTimingInfo::total_time_us)
        .def_readonly("split_time_us", &MaintenanceTimingInfo::split_time_us)
        .def_readonly("delete_time_us", &MaintenanceTimingInfo::delete_time_us)
        .def_readonly("split_refine_time_us", &MaintenanceTimingInfo::split_refine_time_us)
        .def_readonly("delete_refine_time_us", &MaintenanceTimingInfo::delete_refine_time_us)
        .def_readonly("n_splits", &MaintenanceTimingInfo::n_splits)
        .def_readonly("n_deletes", &MaintenanceTimingInfo::n_deletes);

    py::class_<BuildTimingInfo, std::shared_ptr<BuildTimingInfo>>(m, "BuildTimingInfo")
        .def_readonly("n_vectors", &BuildTimingInfo::n_vectors)
        .def_readonly("n_clusters", &BuildTimingInfo::n_clusters)
        .def_readonly("d", &BuildTimingInfo::d)
        .def_readonly("train_time_us", &BuildTimingInfo::train_time_us)
        .def_readonly("assign_time_us", &BuildTimingInfo::assign_time_us)
        .def_readonly("total_time_us", &BuildTimingInfo::total_time_us);

    py::class_<ModifyTimingInfo, std::shared_ptr<ModifyTimingInfo>>(m, "ModifyTimingInfo")
        .def_readonly("n_vectors", &ModifyTimingInfo::n_vectors)
        .def_readonly("modify_count", &ModifyTimingInfo::n_vectors)  // alias, wrap.cpp:264
        .def_readonly("input_validation_time_us", &ModifyTimingInfo::input_validation_time_us)
        .def_readonly("find_partition_time_us", &ModifyTimingInfo::find_partition_time_us)
        .def_readonly("modify_time_us", &ModifyTimingInfo::modify_time_us)
        .def_readonly("maintenance_time_us", &ModifyTimingInfo::maintenance_time_us);

    py::class_<SearchTimingInfo, std::shared_ptr<SearchTimingInfo>>(m, "SearchTimingInfo")
        .def_readwrite("n_queries", &SearchTimingInfo::n_queries)
        .def_readwrite("n_clusters", &SearchTimingInfo::n_clusters)
        .def_readwrite("partitions_scanned", &SearchTimingInfo::partitions_scanned)
        .def_readwrite("search_params", &SearchTimingInfo::search_params)
        .def_readwrite("parent_info", &SearchTimingInfo::parent_info)
        .def_readwrite("buffer_init_time_ns", &SearchTimingInfo::buffer_init_time_ns)
        .def_readwrite("job_enqueue_time_ns", &SearchTimingInfo::job_enqueue_time_ns)
        .def_readwrite("boundary_distance_time_ns", &SearchTimingInfo::boundary_distance_time_ns)
        .def_readwrite("job_wait_time_ns", &SearchTimingInfo::job_wait_time_ns)
        .def_readwrite("result_aggregate_time_ns", &SearchTimingInfo::result_aggregate_time_ns)
        .def_readwrite("total_time_ns", &SearchTimingInfo::total_time_ns);

    py::class_<SearchResult, std::shared_ptr<SearchResult>>(m, "SearchResult")
        .def_readwrite("ids", &SearchResult::ids)
        .def_readwrite("distances", &SearchResult::distances)
        .def_readwrite("timing_info", &SearchResult::timing_info);
}
