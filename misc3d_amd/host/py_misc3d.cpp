// py_misc3d.cpp -- pybind11 module with the reference's python API for the RANSAC hot path
// (python/py_misc3d.cpp:25-62 module wiring, python/py_common.cpp:11-78, python/py_registration.cpp:11-107,
// python/py_segmentation.cpp:87-96), implemented on the host classes of include/misc3d/** which call the
// C ABI (include/misc3d_amd.h) -> HIP kernels.
//
// Same function names, positional order, defaults and return container types as the reference.
// Keyword-only extras that default to the reference's behaviour: seed=None (std::random_device),
// device=0, confidence=0.999 (registration).  Open3D is not a dependency: `pc` may be an (N,3) float64
// array, a (points, normals) tuple, or any object exposing .points / .normals (an
// open3d.geometry.PointCloud works through numpy.asarray).
#include <pybind11/numpy.h>
#include <array>
#include <tuple>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstring>
#include <mutex>
#include <optional>

#include <misc3d/common/ransac.h>
#include <misc3d/logging.h>
#include <misc3d/registration/correspondence_matching.h>
#include <misc3d/registration/transform_estimation.h>
#include <misc3d/segmentation/iterative_plane_segmentation.h>

namespace py = pybind11;
using arr_d = py::array_t<double, py::array::c_style | py::array::forcecast>;

namespace {

struct HostCloud {
    arr_d points, normals;
    bool has_normals = false;
    misc3d::CloudView view() const {
        return misc3d::CloudView(points.size() ? points.data() : nullptr, has_normals ? normals.data() : nullptr,
                                 (size_t)points.shape(0));
    }
};

arr_d as_nx3(const py::object& o, const char* what) {
    py::object np = py::module_::import("numpy");
    arr_d a = arr_d::ensure(np.attr("asarray")(o, py::arg("dtype") = "float64"));
    if (!a) throw py::type_error(std::string(what) + ": cannot convert to a float64 array");
    if (a.ndim() == 2 && a.shape(1) == 3) return a;
    if (a.ndim() == 2 && a.shape(0) == 0) return arr_d(std::vector<py::ssize_t>{0, 3});
    if (a.ndim() == 1 && a.shape(0) == 0) return arr_d(std::vector<py::ssize_t>{0, 3});
    throw py::value_error(std::string(what) + ": expected an (N, 3) array");
}

HostCloud extract_cloud(const py::object& pc) {
    HostCloud c;
    if (py::hasattr(pc, "points")) {  // open3d.geometry.PointCloud or a duck-typed equivalent
        c.points = as_nx3(pc.attr("points"), "pc.points");
        if (py::hasattr(pc, "normals")) {
            py::object n = pc.attr("normals");
            if (!n.is_none()) {
                arr_d nn = as_nx3(n, "pc.normals");
                if (nn.shape(0) == c.points.shape(0) && nn.shape(0) > 0) {
                    c.normals = nn;
                    c.has_normals = true;
                }
            }
        }
    } else if (py::isinstance<py::tuple>(pc) && py::len(pc) == 2) {
        py::tuple t = pc.cast<py::tuple>();
        c.points = as_nx3(t[0], "points");
        if (!t[1].is_none()) {
            c.normals = as_nx3(t[1], "normals");
            if (c.normals.shape(0) != c.points.shape(0)) throw py::value_error("points and normals differ in length");
            c.has_normals = c.normals.shape(0) > 0;
        }
    } else {
        c.points = as_nx3(pc, "pc");
    }
    return c;
}

py::array_t<double> to_array(const std::vector<double>& v) {
    py::array_t<double> a((py::ssize_t)v.size());
    if (!v.empty()) std::memcpy(a.mutable_data(), v.data(), sizeof(double) * v.size());
    return a;
}
py::array_t<double> to_mat4(const misc3d::Matrix4d& T) {
    py::array_t<double> a(std::vector<py::ssize_t>{4, 4});
    std::memcpy(a.mutable_data(), T.data(), sizeof(double) * 16);
    return a;
}

// Page-locked result blocks for calls that return hundreds of megabytes (segment_plane_iterative's cluster clouds): taken
// from / given back to a pool of at most TWO free blocks and kPoolBudget bytes (a loop that rebinds its result keeps one
// generation alive while the next is produced), so a steady caller pins nothing after its second call and an idle process
// holds at most that much (it was four blocks, ~900 MB after one 10 M-point room); release_host_scratch() empties
// the pool.  Never destroyed (capsules may outlive the module at exit).
struct PinnedPool {
    struct Block {
        void* p;
        size_t bytes;
    };
    static constexpr size_t kPoolBudget = (size_t)512 << 20;
    struct State {
        std::mutex mu;
        std::vector<Block> free_blocks, live;
    };
    static void release_free() {
        State& st = state();
        std::vector<Block> drop;
        {
            std::lock_guard<std::mutex> lock(st.mu);
            drop.swap(st.free_blocks);
        }
        for (const Block& b : drop) m3d_host_free(b.p);
    }
    static State& state() {
        static State* s = new State();
        return *s;
    }
    static void* take(size_t bytes) {
        State& st = state();
        {
            std::lock_guard<std::mutex> lock(st.mu);
            for (size_t i = 0; i < st.free_blocks.size(); ++i) {
                const Block b = st.free_blocks[i];
                if (b.bytes >= bytes && b.bytes <= 2 * bytes + ((size_t)1 << 20)) {
                    st.free_blocks.erase(st.free_blocks.begin() + (long)i);
                    st.live.push_back(b);
                    return b.p;
                }
            }
        }
        void* p = m3d_host_alloc(bytes);
        if (p) {
            std::lock_guard<std::mutex> lock(st.mu);
            st.live.push_back({p, bytes});
        }
        return p;
    }
    static void give(void* p) {
        if (!p) return;
        State& st = state();
        Block b{p, 0};
        void* drop = nullptr;
        {
            std::lock_guard<std::mutex> lock(st.mu);
            for (size_t i = 0; i < st.live.size(); ++i)
                if (st.live[i].p == p) {
                    b = st.live[i];
                    st.live.erase(st.live.begin() + (long)i);
                    break;
                }
            size_t held = b.bytes;
            for (const Block& f : st.free_blocks) held += f.bytes;
            if (st.free_blocks.size() < 2 && held <= kPoolBudget) {   // (the two result arrays of one generation)
                st.free_blocks.push_back(b);
            } else {
                drop = p;
            }
        }
        if (drop) m3d_host_free(drop);
    }
};

// FitPlane / FitSphere / FitCylinder, python/py_common.cpp:11-67: construct, SetMaxIteration, SetProbability,
// SetPointCloud, FitModel -- one call of the one-shot C entry point (m3d_fit_plane / _sphere / _cylinder: the same checks in
// the same order, and no resident cloud to keep: a python call fits once).  as_arrays (keyword-only extra): the inlier
// indices as an int64 array instead of the reference's list[int] (half a million Python ints cost ~15 ms to build).
template <int KIND>
py::tuple fit_impl(const py::object& pc, double threshold, size_t max_iteration, double probability,
                   const std::optional<uint64_t>& seed, int device, bool as_arrays) {
    HostCloud c = extract_cloud(pc);
    if (KIND == M3D_CYLINDER && !c.has_normals) misc3d::LogError("Fit cylinder requires normals.");  // py_common.cpp:50-52
    const misc3d::CloudView v = c.view();
    std::vector<double> params(KIND == M3D_CYLINDER ? 7 : 4, 0.0);
    misc3d::detail::PinnedScratch& scratch = misc3d::detail::host_scratch(0);   // (bounded, releasable: geometry.h)
    std::vector<size_t> pageable;
    size_t* idx = nullptr;
    size_t ni = 0;
    m3d_stats st{};
    int rc;
    {
        py::gil_scoped_release nogil;  // the reference holds the GIL; releasing it is unobservable
        idx = static_cast<size_t*>(scratch.get(sizeof(size_t) * (v.n ? v.n : 1)));
        if (!idx) {   // (pinning failed: any host buffer will do)
            pageable.resize(v.n ? v.n : 1);
            idx = pageable.data();
        }
        const uint64_t sd = seed ? *seed : 0;
        if (KIND == M3D_CYLINDER)
            rc = m3d_fit_cylinder(v.xyz, v.normals, v.n, threshold, max_iteration, probability, seed ? &sd : nullptr, device,
                                  params.data(), idx, &ni, &st);
        else
            rc = (KIND == M3D_PLANE ? m3d_fit_plane : m3d_fit_sphere)(v.xyz, v.n, threshold, max_iteration, probability,
                                                                      seed ? &sd : nullptr, device, params.data(), idx, &ni, &st);
    }
    misc3d::CheckStatus(rc);   // "[Misc3D Error] ..." with the reference's texts (probability, lack of points)
    char buf[160];             // ransac.h:616-619
    std::snprintf(buf, sizeof(buf), "Find best model with %g%% inliers and run %llu iterations", st.fitness * 100,
                  (unsigned long long)st.count);
    misc3d::LogInfo(buf);
    if (rc != M3D_OK) params.assign(4, 0.0);  // py_common.cpp:21-23,39-41,61-63 (size 4 for the cylinder too)
    if (as_arrays) {
        py::array_t<int64_t> a((py::ssize_t)ni);
        if (ni) std::memcpy(a.mutable_data(), idx, sizeof(size_t) * ni);
        return py::make_tuple(to_array(params), a);
    }
    return py::make_tuple(to_array(params), std::vector<size_t>(idx, idx + ni));
}

arr_d as_descriptor_matrix(const py::object& f) {
    // open3d Feature (attribute `data`, dim x N) or an ndarray of shape (dim, N)
    py::object np = py::module_::import("numpy");
    py::object src = py::hasattr(f, "data") && !py::isinstance<py::array>(f) ? py::object(f.attr("data")) : f;
    // (a plain py::array here: arr_d would force a C-ordered copy of the (dim, N) array first -- a strided 53 MB
    //  transposition for 200 k FPFH descriptors -- only for the next line to transpose it back)
    py::array a = py::array::ensure(np.attr("asarray")(src, py::arg("dtype") = "float64"));
    if (!a || a.ndim() != 2) throw py::value_error("descriptors: expected a (dim, N) array or an object with .data");
    // Eigen dim x N column-major == N x dim row-major: a Fortran-ordered (dim, N) array -- what Open3D's Feature.data is, and
    // what `descriptors_of_shape_N_dim.T` is -- passes through without a copy
    return arr_d::ensure(np.attr("ascontiguousarray")(a.attr("T")));
}

}  // namespace

// index lists of the reference's API (std::vector<size_t> <-> python list): a numpy integer array is taken by one memcpy
// instead of an element-by-element conversion (94 k correspondences: 2.7 ms of a 4.1 ms compute_transformation_ransac call)
static std::vector<size_t> to_index_vector(const py::handle& o) {
    // (an ndarray must hold integers, and no negative ones: forcecast would truncate floats and wrap -1 to 2^64 - 1 --
    // ADVICE r3; the reference's vector<size_t> conversion raises for both)
    if (py::isinstance<py::array>(o)) {
        const py::array arr = py::reinterpret_borrow<py::array>(o);
        const char kind = arr.dtype().kind();
        if (kind != 'i' && kind != 'u') throw py::type_error("correspondence indices must be integers");
        auto a = py::array_t<int64_t, py::array::c_style | py::array::forcecast>::ensure(arr);
        if (a && a.ndim() == 1) {
            std::vector<size_t> v((size_t)a.shape(0));
            static_assert(sizeof(size_t) == sizeof(int64_t), "size_t is 64 bits on this platform");
            const int64_t* src = a.data();
            for (size_t i = 0; i < v.size(); ++i) {
                if (src[i] < 0) throw py::value_error("correspondence indices must not be negative");
                v[i] = (size_t)src[i];
            }
            return v;
        }
    }
    return py::cast<std::vector<size_t>>(o);
}
static py::array_t<int64_t> index_array(const std::vector<size_t>& v) {
    py::array_t<int64_t> a((py::ssize_t)v.size());
    if (!v.empty()) std::memcpy(a.mutable_data(), v.data(), sizeof(size_t) * v.size());
    return a;
}

PYBIND11_MODULE(_py_misc3d, m) {
    m.doc() = "MI355X-native Misc3D RANSAC hot path (fit_plane/sphere/cylinder, segment_plane_iterative, "
              "compute_transformation_*, match_correspondence)";

    // ---- common (python/py_common.cpp:69-78)
    py::module mc = m.def_submodule("common");
    // EstimateNormalsFromMap, python/py_common.cpp:79-89.  The reference sets pc.normals_ and returns pc; for an
    // object with a writable `.normals` (open3d PointCloud) the same happens, for a plain array the (N, 3)
    // normals are returned.
    mc.def(
        "estimate_normals",
        [](py::object pc, std::tuple<int, int> shape, int k, std::array<double, 3> view_point, int device) -> py::object {
            HostCloud c = extract_cloud(pc);
            const int w = std::get<0>(shape), h = std::get<1>(shape);
            const size_t num = (size_t)c.points.shape(0);
            if (w < 0 || h < 0 || k < 0 || num != (size_t)w * (size_t)h)
                misc3d::LogError("The point cloud size is not equal to given point map size.");
            arr_d normals(std::vector<py::ssize_t>{(py::ssize_t)num, 3});
            int rc;
            {
                py::gil_scoped_release nogil;
                rc = m3d_normals_from_map(num ? c.points.data() : nullptr, (uint32_t)w, (uint32_t)h, (uint32_t)k,
                                          view_point.data(), device, num ? normals.mutable_data() : nullptr, nullptr);
            }
            if (rc < 0) misc3d::LogError(m3d_last_error());
            if (py::hasattr(pc, "points") && py::hasattr(pc, "normals")) {
                py::object value = normals;
                try {   // open3d wants a Vector3dVector
                    value = py::module_::import("open3d").attr("utility").attr("Vector3dVector")(normals);
                } catch (py::error_already_set&) {
                }
                py::setattr(pc, "normals", value);
                return pc;
            }
            return std::move(normals);
        },
        "Estimate normals from pointmap structure", py::arg("pc"), py::arg("shape"), py::arg("k") = 5,
        py::arg("view_point") = std::array<double, 3>{0, 0, 0}, py::kw_only(), py::arg("device") = 0);
    mc.def(
        "fit_plane",
        [](const py::object& pc, double threshold, size_t max_iteration, double probability,
           std::optional<uint64_t> seed, int device, bool as_arrays) {
            return fit_impl<M3D_PLANE>(pc, threshold, max_iteration, probability, seed, device, as_arrays);
        },
        "Fit a plane from point clouds", py::arg("pc"), py::arg("threshold") = 0.01,
        py::arg("max_iteration") = 1000, py::arg("probability") = 0.9999, py::kw_only(),
        py::arg("seed") = py::none(), py::arg("device") = 0, py::arg("as_arrays") = false);
    mc.def(
        "fit_sphere",
        [](const py::object& pc, double threshold, size_t max_iteration, double probability,
           std::optional<uint64_t> seed, int device, bool as_arrays) {
            return fit_impl<M3D_SPHERE>(pc, threshold, max_iteration, probability, seed, device, as_arrays);
        },
        "Fit a sphere from point clouds", py::arg("pc"), py::arg("threshold") = 0.01,
        py::arg("max_iteration") = 1000, py::arg("probability") = 0.9999, py::kw_only(),
        py::arg("seed") = py::none(), py::arg("device") = 0, py::arg("as_arrays") = false);
    mc.def(
        "fit_cylinder",
        [](const py::object& pc, double threshold, size_t max_iteration, double probability,
           std::optional<uint64_t> seed, int device, bool as_arrays) {
            return fit_impl<M3D_CYLINDER>(pc, threshold, max_iteration, probability, seed, device, as_arrays);
        },
        "Fit a cylinder from point clouds", py::arg("pc"), py::arg("threshold") = 0.01,
        py::arg("max_iteration") = 1000, py::arg("probability") = 0.9999, py::kw_only(),
        py::arg("seed") = py::none(), py::arg("device") = 0, py::arg("as_arrays") = false);

    // ---- segmentation (python/py_segmentation.cpp:87-96)
    py::module ms = m.def_submodule("segmentation");
    ms.def(
        "segment_plane_iterative",
        [](const py::object& pcd, double threshold, int max_iteration, double min_ratio,
           std::optional<uint64_t> seed, int device, bool return_indices) {
            // One call of m3d_segment_plane_iterative_clouds (what SegmentPlaneIterativeIndexed makes, without the
            // PlaneCluster copies in between): the clusters' points arrive gathered by the device in ONE (total, 3) array
            // and each cluster is a view of its rows; return_indices: its rows of the one int64 index array.
            HostCloud c = extract_cloud(pcd);
            const misc3d::CloudView v = c.view();
            py::list out;
            if (v.n < 3) {  // iterative_plane_segmentation.cpp:13-17
                misc3d::LogWarning("Point cloud size has less than 3.");
                return out;
            }
            const size_t max_clusters = 4096;
            std::vector<double> planes(4 * max_clusters);
            std::vector<size_t> offsets(max_clusters + 1);
            static_assert(sizeof(size_t) == sizeof(int64_t), "index arrays are 64-bit");
            // Large clouds: the two result arrays are PAGE-LOCKED blocks of a small pool (PinnedPool) -- the library's
            // kernels store the index lists straight into them and the 228 MB of cluster points of a 10 M-point room
            // arrive at the host link's rate (4 ms) instead of through staged copies into fresh pageable pages (25 ms);
            // a block goes back to the pool when the last view of it dies.
            const size_t pts_bytes = sizeof(double) * 3 * v.n, idx_bytes = sizeof(size_t) * v.n;
            const bool pinned = pts_bytes >= ((size_t)4 << 20);
            void* pts_block = pinned ? PinnedPool::take(pts_bytes) : nullptr;
            void* idx_block = pinned && pts_block ? PinnedPool::take(idx_bytes) : nullptr;
            if (pts_block && !idx_block) {
                PinnedPool::give(pts_block);
                pts_block = nullptr;
            }
            py::array_t<int64_t> indices_pageable(pts_block ? (py::ssize_t)0 : (py::ssize_t)v.n);
            arr_d points_pageable(std::vector<py::ssize_t>{pts_block ? (py::ssize_t)0 : (py::ssize_t)v.n, 3});
            size_t* idx_ptr = pts_block ? static_cast<size_t*>(idx_block) : reinterpret_cast<size_t*>(indices_pageable.mutable_data());
            double* pts_ptr = pts_block ? static_cast<double*>(pts_block) : points_pageable.mutable_data();
            size_t k = 0;
            int status;
            {
                py::gil_scoped_release nogil;
                uint64_t s = seed ? *seed : 0;
                status = m3d_segment_plane_iterative_clouds(v.xyz, v.n, threshold, max_iteration, min_ratio,
                                                            seed ? &s : nullptr, device, max_clusters, planes.data(),
                                                            offsets.data(), idx_ptr, pts_ptr, &k);
            }
            py::object points = points_pageable, indices = indices_pageable;
            if (pts_block) {   // (wrapped before anything can throw: the capsules own the blocks from here on)
                const py::ssize_t total = status >= 0 ? (py::ssize_t)offsets[k] : 0;
                points = py::array_t<double>(std::vector<py::ssize_t>{total, 3}, std::vector<py::ssize_t>{24, 8}, pts_ptr,
                                             py::capsule(pts_block, [](void* p) { PinnedPool::give(p); }));
                indices = py::array_t<int64_t>(std::vector<py::ssize_t>{total}, std::vector<py::ssize_t>{8},
                                               reinterpret_cast<int64_t*>(idx_ptr),
                                               py::capsule(idx_block, [](void* p) { PinnedPool::give(p); }));
            }
            if (misc3d::CheckStatus(status) == 2)
                misc3d::LogWarning("segment_plane_iterative: a round found no inlier; stopping early");
            py::object o3d = py::none();
            if (py::hasattr(pcd, "points")) {
                try {
                    o3d = py::module_::import("open3d");
                } catch (py::error_already_set&) {
                    o3d = py::none();
                }
            }
            for (size_t cix = 0; cix < k; ++cix) {
                py::array_t<double> plane(4);
                std::memcpy(plane.mutable_data(), &planes[4 * cix], sizeof(double) * 4);
                const py::slice rows((py::ssize_t)offsets[cix], (py::ssize_t)offsets[cix + 1], 1);
                py::object cluster = points[rows];   // (a view: keeps the block alive)
                if (!o3d.is_none())  // list[(ndarray(4), open3d PointCloud)] like the reference
                    cluster = o3d.attr("geometry").attr("PointCloud")(o3d.attr("utility").attr("Vector3dVector")(cluster));
                if (return_indices)
                    out.append(py::make_tuple(plane, cluster, indices[rows]));
                else
                    out.append(py::make_tuple(plane, cluster));
            }
            return out;
        },
        "Segment plane iteratively using RANSAC plane fitting", py::arg("pcd"), py::arg("threshold"),
        py::arg("max_iteration") = 100, py::arg("min_ratio") = 0.05, py::kw_only(), py::arg("seed") = py::none(),
        py::arg("device") = 0, py::arg("return_indices") = false);

    // ---- registration (python/py_registration.cpp:11-107)
    py::module mr = m.def_submodule("registration");
    mr.def(
        "compute_transformation_least_square",
        [](const py::object& src, const py::object& dst, bool scaling, int device) {
            auto pts = [](const py::object& o) {
                if (py::hasattr(o, "points")) return extract_cloud(o).points;
                py::object np = py::module_::import("numpy");
                arr_d a = arr_d::ensure(np.attr("asarray")(o, py::arg("dtype") = "float64"));
                // the ndarray overload is documented as (n, 3) but typed Matrix3Xd (py_registration.cpp:22-27):
                // accept both orientations
                if (a && a.ndim() == 2 && a.shape(1) != 3 && a.shape(0) == 3)
                    return arr_d::ensure(np.attr("ascontiguousarray")(a.attr("T")));
                return as_nx3(o, "points");
            };
            arr_d s = pts(src), d = pts(dst);
            misc3d::Matrix4d T;
            {
                py::gil_scoped_release nogil;
                misc3d::registration::LeastSquareSolver solver(scaling, device);
                T = solver.Solve(misc3d::CloudView(s.data(), nullptr, (size_t)s.shape(0)),
                                 misc3d::CloudView(d.data(), nullptr, (size_t)d.shape(0)));
            }
            return to_mat4(T);
        },
        "Compute 3D transformation from corresponding point clouds using Least-Square method", py::arg("src"),
        py::arg("dst"), py::arg("scaling") = false, py::kw_only(), py::arg("device") = 0);
    mr.def(
        "compute_transformation_teaser",
        [](const py::object&, const py::object&, double) -> py::object {
            // (through LogError like every other failure of this module: "[Misc3D Error] ..." as a RuntimeError)
            misc3d::LogError(
                "compute_transformation_teaser (TEASER++, CPU graph solver) is outside the MI355X-accelerated hot path of "
                "this build; use compute_transformation_ransac or compute_transformation_least_square");
        },
        py::arg("src"), py::arg("dst"), py::arg("noise_bound") = 0.01);
    mr.def(
        "compute_transformation_ransac",
        [](const py::object& src, const py::object& dst, const py::object& corres_obj, double threshold, int max_iter,
           double edge_length_threshold, std::optional<uint64_t> seed, double confidence, int device) {
            HostCloud s = extract_cloud(src), d = extract_cloud(dst);
            // (list, list) as in the reference, or two integer arrays (what match_correspondence(..., as_arrays=True) returns)
            const py::sequence cs = py::reinterpret_borrow<py::sequence>(corres_obj);
            if (py::len(cs) != 2) throw py::value_error("corres must be a pair (indices into src, indices into dst)");
            std::pair<std::vector<size_t>, std::vector<size_t>> corres{to_index_vector(cs[0]), to_index_vector(cs[1])};
            misc3d::Matrix4d T;
            {
                py::gil_scoped_release nogil;
                misc3d::registration::RANSACSolver solver(threshold, max_iter, edge_length_threshold);
                solver.SetConfidence(confidence);
                solver.SetDevice(device);
                if (seed) solver.SetSeed(*seed);
                solver.SetWantStats(false);   // (the reference's function returns the pose and nothing else)
                T = solver.Solve(s.view(), d.view(), corres);
            }
            return to_mat4(T);
        },
        "Compute 3D rigid transformation from corresponding point clouds using RANSAC", py::arg("src"),
        py::arg("dst"), py::arg("corres"), py::arg("threshold") = 0.01, py::arg("max_iter") = 100000,
        py::arg("edge_length_threshold") = 0.9, py::kw_only(), py::arg("seed") = py::none(),
        py::arg("confidence") = 0.999, py::arg("device") = 0);
    py::enum_<misc3d::registration::MatchMethod>(mr, "MatchMethod")
        .value("FLANN", misc3d::registration::MatchMethod::FLANN)
        .value("ANNOY", misc3d::registration::MatchMethod::ANNOY)
        .export_values();
    mr.def(
        "match_correspondence",
        [](const py::object& src, const py::object& dst, const misc3d::registration::MatchMethod& method,
           int n_trees, int device, bool as_arrays) -> py::object {
            arr_d fs = as_descriptor_matrix(src), fd = as_descriptor_matrix(dst);
            if (fs.shape(1) != fd.shape(1)) throw py::value_error("descriptor dimensions differ");
            std::pair<std::vector<size_t>, std::vector<size_t>> res;
            {
                py::gil_scoped_release nogil;
                misc3d::registration::ANNMatcher matcher(method, n_trees);
                matcher.SetDevice(device);
                misc3d::registration::FeatureView a{fs.data(), (int)fs.shape(1), (size_t)fs.shape(0)};
                misc3d::registration::FeatureView b{fd.data(), (int)fd.shape(1), (size_t)fd.shape(0)};
                res = matcher.Match(a, b);
            }
            if (as_arrays) return py::make_tuple(index_array(res.first), index_array(res.second));
            return py::cast(res);   // (list, list), as the reference returns them
        },
        "Match corresponding point clouds (mutual nearest neighbours in descriptor space).  The search is EXACT for both "
        "methods: MatchMethod.ANNOY (the reference's default, an approximate random-projection forest) and n_trees are "
        "accepted for compatibility and give the FLANN result; on real descriptors the reference's ANNOY set can differ "
        "from it by the forest's misses.",
        py::arg("src"),
        py::arg("dst"), py::arg("method") = misc3d::registration::MatchMethod::ANNOY, py::arg("n_trees") = 4,
        py::kw_only(), py::arg("device") = 0, py::arg("as_arrays") = false);

    // ---- logging (python/py_misc3d.cpp:52-62)
    py::enum_<misc3d::VerbosityLevel>(m, "VerbosityLevel", py::arithmetic(), "VerbosityLevel")
        .value("Error", misc3d::VerbosityLevel::Error)
        .value("Warning", misc3d::VerbosityLevel::Warning)
        .value("Info", misc3d::VerbosityLevel::Info)
        .value("Debug", misc3d::VerbosityLevel::Debug)
        .export_values();
    m.def("set_verbosity_level", &misc3d::SetVerbosityLevel, "Set global verbosity level of Misc3D",
          py::arg("verbosity_level"));
    m.def("get_verbosity_level", &misc3d::GetVerbosityLevel, "Get global verbosity level of Misc3D");
    m.def("device_count", []() { return m3d_device_count(); });
    m.def("release_host_scratch", []() {
        misc3d::ReleaseHostScratch();
        PinnedPool::release_free();
    }, "Free the page-locked blocks the calling thread and the result pool keep between calls");
}
