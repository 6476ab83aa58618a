// bindings.cpp -- pybind11 module quake_amd._bindings: the classes, attributes and method names of the reference's
// `quake._bindings` (src/cpp/bindings/wrap.cpp:48-368), bound to the C++ host mirror in quake_index.h.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>
#include <torch/extension.h>

#include <sstream>

#include "quake_index.h"

namespace py = pybind11;
using namespace quake_amd;

namespace {
// the JSON-style one-line summaries of wrap.cpp's __repr__s: Repr().kv("a", 1).kv("b", true).str() -> {"a": 1, "b": true}
// (`trailing` reproduces the ", }" two of the reference's summaries end in)
class Repr {
    std::ostringstream os_;
    bool first_ = true;
    void key(const char *k) {
        os_ << (first_ ? "" : ", ") << '"' << k << "\": ";
        first_ = false;
    }

public:
    template <typename T>
    Repr &kv(const char *k, const T &v) {
        key(k);
        os_ << v;
        return *this;
    }
    Repr &kv(const char *k, bool v) {
        key(k);
        os_ << (v ? "true" : "false");
        return *this;
    }
    Repr &kv(const char *k, const std::string &v) {
        key(k);
        os_ << '"' << v << '"';
        return *this;
    }
    std::string str(bool trailing = false) const { return "{" + os_.str() + (trailing ? ", }" : "}"); }
};
}  // namespace

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
        .def("set_track_hits", &QuakeIndex::set_track_hits, "Record the partitions each query probes, so that maintenance() can act.")
        .def("set_latency_profile", &QuakeIndex::set_latency_profile, "Scan-latency grid of the maintenance cost model from a CSV (reference profile format).")
        .def("validate", &QuakeIndex::validate)
        .def_readonly("parent", &QuakeIndex::parent_)
        .def_readonly("partition_manager", &QuakeIndex::partition_manager_)
        .def_readonly("query_coordinator", &QuakeIndex::query_coordinator_)
        .def_readonly("current_level", &QuakeIndex::current_level_)
        .def("__repr__", [](const QuakeIndex &q) { return Repr().kv("current_level", q.current_level_).str(true); });  // wrap.cpp:122-128

    // the collaborators the reference's own tests reach into (test/cpp/quake_index.cpp:50-54, query_coordinator.cpp:42,96)
    py::class_<PartitionManager, std::shared_ptr<PartitionManager>>(m, "PartitionManager")
        .def("ntotal", &PartitionManager::ntotal)
        .def("nlist", &PartitionManager::nlist)
        .def("d", &PartitionManager::d)
        .def("get_partition_ids", &PartitionManager::get_partition_ids)
        .def("get_partition_sizes", [](PartitionManager &pm, torch::Tensor pids) { return pm.get_partition_sizes(pids); })
        .def("get_ids", &PartitionManager::get_ids)
        .def("validate", &PartitionManager::validate)
        // workers = GPUs: the partition -> worker map (partition_manager.h:170-187)
        .def("distribute_partitions", &PartitionManager::distribute_partitions)
        .def("get_partition_core_id", &PartitionManager::get_partition_core_id)
        .def("set_partition_core_id", &PartitionManager::set_partition_core_id)
        .def("num_workers", &PartitionManager::num_workers, "Members of the device group the partitions are distributed over (0: one store).");

    py::class_<QueryCoordinator, std::shared_ptr<QueryCoordinator>>(m, "QueryCoordinator")
        .def("search", &QueryCoordinator::search)
        .def("scan_partitions", &QueryCoordinator::scan_partitions)
        .def("serial_scan", &QueryCoordinator::serial_scan)
        .def("batched_serial_scan", &QueryCoordinator::batched_serial_scan)
        .def("worker_scan", &QueryCoordinator::worker_scan)
        .def("initialize_workers", &QueryCoordinator::initialize_workers)  // query_coordinator.h:150-160
        .def("shutdown_workers", &QueryCoordinator::shutdown_workers)
        .def_readwrite("device_timing", &QueryCoordinator::device_timing_)  // query_coordinator.h: counters for device-tensor searches
        .def_readonly("workers_initialized", &QueryCoordinator::workers_initialized_)
        .def_readonly("num_workers", &QueryCoordinator::num_workers_);

    // list_scanning.h seam: one query (or a block of queries) against one raw list
    m.def("batched_scan_list", [](torch::Tensor queries, torch::Tensor list_vecs, torch::Tensor list_ids, int k, std::string metric) {
        torch::Tensor q = queries.to(torch::kCPU, torch::kFloat32).contiguous(), v = list_vecs.to(torch::kCPU, torch::kFloat32).contiguous();
        torch::Tensor ids = list_ids.defined() && list_ids.numel() ? list_ids.to(torch::kCPU, torch::kInt64).contiguous() : torch::Tensor();
        const MetricType mt = (MetricType)str_to_metric_type(metric);
        auto bufs = create_buffers((int)q.size(0), k, mt == METRIC_INNER_PRODUCT);
        batched_scan_list(q.data_ptr<float>(), v.data_ptr<float>(), ids.defined() ? ids.data_ptr<int64_t>() : nullptr, (int)q.size(0),
                          (int)v.size(0), (int)q.size(1), bufs, mt);
        return buffers_to_tensor(bufs);
    }, py::arg("queries"), py::arg("list_vecs"), py::arg("list_ids"), py::arg("k"), py::arg("metric") = "l2");

    py::class_<IndexBuildParams, std::shared_ptr<IndexBuildParams>>(m, "IndexBuildParams")
        .def(py::init<>())
        .def_readwrite("nlist", &IndexBuildParams::nlist)
        .def_readwrite("niter", &IndexBuildParams::niter)
        .def_readwrite("metric", &IndexBuildParams::metric)
        .def_readwrite("num_workers", &IndexBuildParams::num_workers)
        .def("__repr__", [](const IndexBuildParams &p) {  // wrap.cpp:141-150
            return Repr().kv("nlist", p.nlist).kv("niter", p.niter).kv("metric", p.metric).kv("num_workers", p.num_workers).str();
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
        .def_readwrite("aps_flush_period_us", &SearchParams::aps_flush_period_us)
        .def("__repr__", [](const SearchParams &p) {  // wrap.cpp:173-186
            return Repr().kv("k", p.k).kv("nprobe", p.nprobe).kv("recall_target", p.recall_target).kv("batched_scan", p.batched_scan)
                .kv("use_precomputed", p.use_precomputed).kv("initial_search_fraction", p.initial_search_fraction)
                .kv("recompute_threshold", p.recompute_threshold).kv("aps_flush_period_us", p.aps_flush_period_us).str();
        });

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
        .def_readwrite("split_threshold_ns", &MaintenancePolicyParams::split_threshold_ns)
        .def_readwrite("split_after_delete_rejection", &MaintenancePolicyParams::split_after_delete_rejection)  // extension, common.h
        .def("__repr__", [](const MaintenancePolicyParams &p) {  // wrap.cpp:211-226
            return Repr().kv("maintenance_policy", p.maintenance_policy).kv("window_size", p.window_size)
                .kv("refinement_radius", p.refinement_radius).kv("refinement_iterations", p.refinement_iterations)
                .kv("min_partition_size", p.min_partition_size).kv("alpha", p.alpha)
                .kv("enable_split_rejection", p.enable_split_rejection).kv("enable_delete_rejection", p.enable_delete_rejection)
                .kv("delete_threshold_ns", p.delete_threshold_ns).kv("split_threshold_ns", p.split_threshold_ns).str(true);
        });

    py::class_<MaintenanceTimingInfo, std::shared_ptr<MaintenanceTimingInfo>>(m, "MaintenanceTimingInfo")
        .def_readonly("total_time_us", &MaintenanceTimingInfo::total_time_us)
        .def_readonly("split_time_us", &MaintenanceTimingInfo::split_time_us)
        .def_readonly("delete_time_us", &MaintenanceTimingInfo::delete_time_us)
        .def_readonly("split_refine_time_us", &MaintenanceTimingInfo::split_refine_time_us)
        .def_readonly("delete_refine_time_us", &MaintenanceTimingInfo::delete_refine_time_us)
        .def_readonly("n_splits", &MaintenanceTimingInfo::n_splits)
        .def_readonly("n_deletes", &MaintenanceTimingInfo::n_deletes)
        .def("__repr__", [](const MaintenanceTimingInfo &t) {  // wrap.cpp:244-256
            return Repr().kv("total_time_us", t.total_time_us).kv("split_time_us", t.split_time_us).kv("delete_time_us", t.delete_time_us)
                .kv("split_refine_time_us", t.split_refine_time_us).kv("delete_refine_time_us", t.delete_refine_time_us)
                .kv("n_splits", t.n_splits).kv("n_deletes", t.n_deletes).str();
        });

    py::class_<BuildTimingInfo, std::shared_ptr<BuildTimingInfo>>(m, "BuildTimingInfo")
        .def_readonly("n_vectors", &BuildTimingInfo::n_vectors)
        .def_readonly("n_clusters", &BuildTimingInfo::n_clusters)
        .def_readonly("d", &BuildTimingInfo::d)
        .def_readonly("train_time_us", &BuildTimingInfo::train_time_us)
        .def_readonly("assign_time_us", &BuildTimingInfo::assign_time_us)
        .def_readonly("total_time_us", &BuildTimingInfo::total_time_us)
        .def_readonly("code_size", &BuildTimingInfo::code_size)        // wrap.cpp:332-335 (PQ fields: -1, no PQ on this path)
        .def_readonly("n_codebooks", &BuildTimingInfo::num_codebooks)
        .def("__repr__", [](const BuildTimingInfo &b) {  // wrap.cpp:338-350
            return Repr().kv("total_time_us", b.total_time_us).kv("assign_time_us", b.assign_time_us).kv("train_time_us", b.train_time_us)
                .kv("d", b.d).kv("code_size", b.code_size).kv("n_codebooks", b.num_codebooks).kv("n_vectors", b.n_vectors).str();
        });

    py::class_<ModifyTimingInfo, std::shared_ptr<ModifyTimingInfo>>(m, "ModifyTimingInfo")
        .def_readonly("n_vectors", &ModifyTimingInfo::n_vectors)
        .def_readonly("modify_count", &ModifyTimingInfo::n_vectors)  // alias, wrap.cpp:264
        .def_readonly("input_validation_time_us", &ModifyTimingInfo::input_validation_time_us)
        .def_readonly("find_partition_time_us", &ModifyTimingInfo::find_partition_time_us)
        .def_readonly("modify_time_us", &ModifyTimingInfo::modify_time_us)
        .def_readonly("maintenance_time_us", &ModifyTimingInfo::maintenance_time_us)
        .def("__repr__", [](const ModifyTimingInfo &t) {  // wrap.cpp:267-276
            return Repr().kv("modify_count", t.n_vectors).kv("input_validation_time_us", t.input_validation_time_us)
                .kv("modify_time_us", t.modify_time_us).kv("find_partition_time_us", t.find_partition_time_us).str();
        });

    py::class_<SearchTimingInfo, std::shared_ptr<SearchTimingInfo>>(m, "SearchTimingInfo")
        .def(py::init<>())
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
        .def_readwrite("total_time_ns", &SearchTimingInfo::total_time_ns)
        .def("__repr__", [](const SearchTimingInfo &t) {  // wrap.cpp:303-320
            Repr r;
            r.kv("total_time_ns", t.total_time_ns).kv("buffer_init_time_ns", t.buffer_init_time_ns)
                .kv("job_enqueue_time_ns", t.job_enqueue_time_ns).kv("boundary_distance_time_ns", t.boundary_distance_time_ns)
                .kv("job_wait_time_ns", t.job_wait_time_ns).kv("result_aggregate_time_ns", t.result_aggregate_time_ns);
            if (t.parent_info) r.kv("parent_scan_time_ns", t.parent_info->total_time_ns);
            return r.kv("n_queries", t.n_queries).kv("n_clusters", t.n_clusters).kv("partitions_scanned", t.partitions_scanned).str();
        });

    py::class_<SearchResult, std::shared_ptr<SearchResult>>(m, "SearchResult")
        .def(py::init<>())
        .def_readwrite("ids", &SearchResult::ids)
        .def_readwrite("distances", &SearchResult::distances)
        .def_readwrite("timing_info", &SearchResult::timing_info);
}
