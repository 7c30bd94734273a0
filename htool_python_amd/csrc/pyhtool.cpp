// pyhtool.cpp -- thin pybind11 module "Htool" over the C ABI of libhtool_mi355x.so.
//
// Mirrors the Python surface the reference registers in src/htool/main.cpp:40-112 for the H-matrix
// build + product path (class names, argument names, defaults, error messages); every method only
// converts arguments and forwards to include/htool_mi355x.h.  No arithmetic happens here.
#include <pybind11/functional.h>
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <complex>
#include <cstring>
#include <map>
#include <memory>
#include <optional>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/htool_mi355x.h"

namespace py = pybind11;
using namespace pybind11::literals;

static void check(int rc) {
    if (rc != 0) throw std::runtime_error(htool_last_error());
}

// ---- logging: C sink -> logging.getLogger("Htool") (src/htool/misc/logger.hpp:13-33) ------------
static void python_log_sink(int level, const char *message) {
    py::gil_scoped_acquire gil;
    py::object logger = py::module::import("logging").attr("getLogger")("Htool");
    switch (level) {
    case 0: logger.attr("critical")(message); break;
    case 1: logger.attr("error")(message); break;
    case 2: logger.attr("warning")(message); break;
    case 3: logger.attr("debug")(message); break;
    default: logger.attr("info")(message); break;
    }
}

// ---- clusters -------------------------------------------------------------------------------------
struct ClusterRoot {
    htool_cluster *root = nullptr;
    ~ClusterRoot() { htool_cluster_destroy(root); }
};
struct PyCluster {
    std::shared_ptr<ClusterRoot> owner;
    const htool_cluster *node = nullptr;
};

struct PyPartitioning {
    int strategy = HTOOL_PCA_REGULAR;
    virtual ~PyPartitioning() = default;
};
template <int S>
struct PyPartitioningT : PyPartitioning {
    PyPartitioningT() { strategy = S; }
};

struct PyClusterTreeBuilder {
    int max_leaf = 10;
    int strategy = HTOOL_PCA_REGULAR;
    typedef py::array_t<double, py::array::f_style | py::array::forcecast> coords_t;
    typedef py::array_t<int, py::array::f_style | py::array::forcecast> part_t;

    PyCluster create(coords_t coordinates, int number_of_children, int size_of_partition, const int *partition, bool local,
                     std::optional<coords_t> radii, std::optional<coords_t> weights) {
        if (coordinates.ndim() != 2) throw std::runtime_error("coordinates must be a (dimension, number_of_points) array");
        auto owner = std::make_shared<ClusterRoot>();
        // a (d, N) Fortran-ordered array is point-major in memory (cluster_tree_builder.hpp:19-23)
        check(htool_cluster_create(coordinates.data(), (int)coordinates.shape(1), (int)coordinates.shape(0), radii ? radii->data() : nullptr,
                                   weights ? weights->data() : nullptr, number_of_children, size_of_partition, partition, local ? 1 : 0, max_leaf, strategy,
                                   &owner->root));
        return PyCluster{owner, owner->root};
    }
};

// ---- generators -----------------------------------------------------------------------------------
template <typename T>
struct PyIGenerator {
    virtual ~PyIGenerator() = default;
    htool_generator *handle = nullptr; // created lazily
    virtual htool_generator *get() = 0;
};

template <typename T>
struct PyVirtualGenerator : PyIGenerator<T> {
    PyVirtualGenerator() {}
    PyVirtualGenerator(const py::array_t<int> &, const py::array_t<int> &) {}
    ~PyVirtualGenerator() override { if (this->handle) htool_generator_destroy(this->handle); }
    virtual void build_submatrix(const py::array_t<int> &J, const py::array_t<int> &K, py::array_t<T, py::array::f_style> &mat) const = 0;

    // C callback -> Python build_submatrix; numpy views over the library's buffers, no copies
    // (src/htool/hmatrix/interfaces/virtual_generator.hpp:16-25)
    static void trampoline(void *ctx, int M, int N, const int *rows, const int *cols, void *out) {
        if ((long)M * N <= 0) return;
        auto *self = static_cast<PyVirtualGenerator<T> *>(ctx);
        py::array_t<T, py::array::f_style> mat(std::array<py::ssize_t, 2>{M, N}, (T *)out, py::capsule(out, [](void *) {}));
        py::array_t<int> py_rows(std::array<py::ssize_t, 1>{M}, rows, py::capsule(rows, [](void *) {}));
        py::array_t<int> py_cols(std::array<py::ssize_t, 1>{N}, cols, py::capsule(cols, [](void *) {}));
        self->build_submatrix(py_rows, py_cols, mat);
    }
    htool_generator *get() override {
        if (!this->handle) check(htool_generator_create_callback(std::is_same<T, double>::value ? 0 : 1, &trampoline, this, &this->handle));
        return this->handle;
    }
};

template <typename T>
struct PyVirtualGeneratorTrampoline : PyVirtualGenerator<T> {
    using PyVirtualGenerator<T>::PyVirtualGenerator;
    void build_submatrix(const py::array_t<int> &J, const py::array_t<int> &K, py::array_t<T, py::array::f_style> &mat) const override {
        PYBIND11_OVERRIDE_PURE(void, PyVirtualGenerator<T>, build_submatrix, J, K, mat);
    }
};

// native generator: kernel evaluated on the device (extension of the reference API; the reference only
// has callback generators).  kind: "inv_delta" | "laplace" | "helmholtz"
template <typename T>
struct PyNativeGenerator : PyIGenerator<T> {
    typedef py::array_t<double, py::array::f_style | py::array::forcecast> coords_t;
    PyNativeGenerator(const std::string &kind, coords_t target_points, coords_t source_points, double param) {
        int k = kind == "inv_delta" ? HTOOL_KERNEL_INV_DELTA : kind == "laplace" ? HTOOL_KERNEL_LAPLACE : kind == "helmholtz" ? HTOOL_KERNEL_HELMHOLTZ : -1;
        if (k < 0) throw std::runtime_error("unknown kernel kind '" + kind + "'");
        if ((k == HTOOL_KERNEL_HELMHOLTZ) != !std::is_same<T, double>::value) throw std::runtime_error("kernel kind does not match the coefficient type of this generator class");
        if (target_points.ndim() != 2 || source_points.ndim() != 2 || target_points.shape(0) != source_points.shape(0)) throw std::runtime_error("points must be (dimension, n) arrays");
        check(htool_generator_create_native(k, (int)target_points.shape(0), target_points.data(), (int)target_points.shape(1), source_points.data(),
                                            (int)source_points.shape(1), param, &this->handle));
    }
    ~PyNativeGenerator() override { if (this->handle) htool_generator_destroy(this->handle); }
    htool_generator *get() override { return this->handle; }
};

// ---- custom low-rank generator (virtual_low_rank_generator.hpp:17-92) ------------------------------
template <typename T>
struct PyVirtualLowRankGenerator {
    typedef typename std::conditional<std::is_same<T, double>::value, double, double>::type real_t;
    mutable std::vector<py::array_t<T, py::array::f_style>> mats_U, mats_V; // owned by Python
    bool allow_copy = true;
    mutable std::vector<T> scratch_U, scratch_V;
    explicit PyVirtualLowRankGenerator(bool allow_copy_ = true) : allow_copy(allow_copy_) {}
    virtual ~PyVirtualLowRankGenerator() = default;
    virtual bool build_low_rank_approximation(const py::array_t<int, py::array::f_style> &rows, const py::array_t<int, py::array::f_style> &cols, double epsilon) const = 0;
    void set_U(py::array_t<T, py::array::f_style> U0) { mats_U.push_back(U0); }
    void set_V(py::array_t<T, py::array::f_style> V0) { mats_V.push_back(V0); }
    void clear_data() {
        mats_U.clear();
        mats_V.clear();
    }
    static int trampoline(void *ctx, int M, int N, const int *rows, const int *cols, double epsilon, const void **U, const void **V, int *rank) {
        auto *self = static_cast<PyVirtualLowRankGenerator<T> *>(ctx);
        py::array_t<int, py::array::f_style> py_rows(std::array<py::ssize_t, 1>{M}, rows, py::capsule(rows, [](void *) {}));
        py::array_t<int, py::array::f_style> py_cols(std::array<py::ssize_t, 1>{N}, cols, py::capsule(cols, [](void *) {}));
        size_t before_u = self->mats_U.size(), before_v = self->mats_V.size();
        bool ok = self->build_low_rank_approximation(py_rows, py_cols, epsilon);
        if (!ok) return 0;
        if (self->mats_U.size() <= before_u || self->mats_V.size() <= before_v) throw std::runtime_error("build_low_rank_approximation returned True without calling set_U and set_V");
        auto &Um = self->mats_U.back();
        auto &Vm = self->mats_V.back();
        if (Um.ndim() != 2 || Vm.ndim() != 2 || Um.shape(0) != M || Vm.shape(1) != N || Um.shape(1) != Vm.shape(0)) throw std::runtime_error("set_U/set_V: expected U (M x r) and V (r x N)");
        *rank = (int)Um.shape(1);
        // allow_copy (virtual_low_rank_generator.hpp:33-38): the factors are copied out and the Python references dropped
        // right away; otherwise (:39-42) the library borrows the Python arrays, which stay alive until clear_data()
        // (htool_build_params.compress_borrows: read once, when the panels are shipped to HBM)
        *U = Um.data();
        *V = Vm.data();
        if (self->allow_copy) {
            self->scratch_U.assign(Um.data(), Um.data() + Um.size());
            self->scratch_V.assign(Vm.data(), Vm.data() + Vm.size());
            *U = self->scratch_U.data();
            *V = self->scratch_V.data();
            self->mats_U.pop_back();
            self->mats_V.pop_back();
        }
        return 1;
    }
};
template <typename T>
struct PyVirtualLowRankGeneratorTrampoline : PyVirtualLowRankGenerator<T> {
    using PyVirtualLowRankGenerator<T>::PyVirtualLowRankGenerator;
    bool build_low_rank_approximation(const py::array_t<int, py::array::f_style> &rows, const py::array_t<int, py::array::f_style> &cols, double epsilon) const override {
        PYBIND11_OVERRIDE_PURE(bool, PyVirtualLowRankGenerator<T>, build_low_rank_approximation, rows, cols, epsilon);
    }
};

// ---- custom dense blocks generator (virtual_dense_blocks_generator.hpp:10-69) -----------------------
template <typename T>
struct PyVirtualDenseBlocksGenerator {
    PyCluster target, source;
    PyVirtualDenseBlocksGenerator(const PyCluster &t, const PyCluster &s) : target(t), source(s) {}
    virtual ~PyVirtualDenseBlocksGenerator() = default;
    virtual void build_dense_blocks(const std::vector<py::array_t<int, py::array::f_style>> &rows, const std::vector<py::array_t<int, py::array::f_style>> &cols,
                                    std::vector<py::array_t<T, py::array::f_style>> &blocks) const = 0;
    static void trampoline(void *ctx, int nb, const int *M, const int *N, const int *row_offsets, const int *col_offsets, void **ptrs) {
        auto *self = static_cast<PyVirtualDenseBlocksGenerator<T> *>(ctx);
        int nt = 0, ns = 0;
        const int *pt = htool_cluster_permutation(self->target.node, &nt), *ps = htool_cluster_permutation(self->source.node, &ns);
        std::vector<py::array_t<T, py::array::f_style>> vec_ptr;
        std::vector<py::array_t<int, py::array::f_style>> rows_ptr, cols_ptr;
        for (int i = 0; i < nb; i++) {
            rows_ptr.emplace_back(std::array<py::ssize_t, 1>{M[i]}, pt + row_offsets[i], py::capsule(pt, [](void *) {}));
            cols_ptr.emplace_back(std::array<py::ssize_t, 1>{N[i]}, ps + col_offsets[i], py::capsule(ps, [](void *) {}));
            vec_ptr.emplace_back(std::array<py::ssize_t, 2>{M[i], N[i]}, (T *)ptrs[i], py::capsule(ptrs[i], [](void *) {}));
        }
        self->build_dense_blocks(rows_ptr, cols_ptr, vec_ptr);
    }
};
template <typename T>
struct PyVirtualDenseBlocksGeneratorTrampoline : PyVirtualDenseBlocksGenerator<T> {
    using PyVirtualDenseBlocksGenerator<T>::PyVirtualDenseBlocksGenerator;
    void build_dense_blocks(const std::vector<py::array_t<int, py::array::f_style>> &rows, const std::vector<py::array_t<int, py::array::f_style>> &cols,
                            std::vector<py::array_t<T, py::array::f_style>> &blocks) const override {
        PYBIND11_OVERRIDE_PURE(void, PyVirtualDenseBlocksGenerator<T>, build_dense_blocks, rows, cols, blocks);
    }
};

// ---- H-matrix -------------------------------------------------------------------------------------
static std::map<std::string, std::string> parse_info(const htool_hmatrix *h, int which) {
    int need = htool_hmatrix_info(h, which, nullptr, 0);
    std::string buf((size_t)need, '\0');
    htool_hmatrix_info(h, which, &buf[0], need);
    std::map<std::string, std::string> out;
    std::istringstream in(buf.c_str());
    std::string line;
    while (std::getline(in, line)) {
        auto eq = line.find('=');
        if (eq != std::string::npos) out[line.substr(0, eq)] = line.substr(eq + 1);
    }
    return out;
}

template <typename T>
struct PyHMatrix {
    htool_hmatrix *h = nullptr;
    bool owned = true;
    PyCluster target, source; // keep the cluster trees alive (the reference stores references)
    PyHMatrix() {}
    PyHMatrix(const PyHMatrix &) = delete;
    PyHMatrix(PyHMatrix &&o) noexcept : h(o.h), owned(o.owned), target(o.target), source(o.source) { o.h = nullptr; }
    ~PyHMatrix() { if (h && owned) htool_hmatrix_destroy(h); }

    py::array_t<T, py::array::f_style> mul(const py::array_t<T, py::array::f_style> &input) const {
        if (input.ndim() != 1) throw std::runtime_error("Wrong dimension for HMatrix-vector product");
        if (input.shape(0) != htool_hmatrix_nb_cols(h)) throw std::runtime_error("Wrong size for HMatrix-vector product");
        py::array_t<T, py::array::f_style> result(htool_hmatrix_nb_rows(h));
        std::fill_n(result.mutable_data(), result.size(), T(0));
        T one(1), zero(0);
        int rc;
        {
            py::gil_scoped_release nogil; // no Python code runs inside a product (the log sink re-acquires the GIL itself)
            rc = htool_hmatrix_matvec(h, 'N', &one, input.data(), &zero, result.mutable_data());
        }
        check(rc);
        return result;
    }
    py::array_t<T, py::array::f_style> matmul(const py::array_t<T, py::array::f_style> &input) const {
        if (input.ndim() != 2) throw std::runtime_error("Wrong dimension for HMatrix-matrix product");
        if (input.shape(0) != htool_hmatrix_nb_cols(h)) throw std::runtime_error("Wrong size for HMatrix-matrix product");
        py::array_t<T, py::array::f_style> result({(py::ssize_t)htool_hmatrix_nb_rows(h), input.shape(1)});
        std::fill_n(result.mutable_data(), result.size(), T(0));
        T one(1), zero(0);
        int rc;
        {
            py::gil_scoped_release nogil;
            rc = htool_hmatrix_matmat(h, 'N', &one, input.data(), (int)input.shape(1), &zero, result.mutable_data());
        }
        check(rc);
        return result;
    }
    // extension: y = H^T x ('T') or H^H x ('C'), one column or several; x has one entry per row of H
    py::array_t<T, py::array::f_style> transposed_mul(const py::array_t<T, py::array::f_style> &input, char trans) const {
        if (input.ndim() != 1 && input.ndim() != 2) throw std::runtime_error("Wrong dimension for transposed HMatrix product");
        if (input.shape(0) != htool_hmatrix_nb_rows(h)) throw std::runtime_error("Wrong size for transposed HMatrix product");
        const int mu = input.ndim() == 1 ? 1 : (int)input.shape(1);
        py::array_t<T, py::array::f_style> result = input.ndim() == 1 ? py::array_t<T, py::array::f_style>(htool_hmatrix_nb_cols(h))
                                                                        : py::array_t<T, py::array::f_style>({(py::ssize_t)htool_hmatrix_nb_cols(h), (py::ssize_t)mu});
        std::fill_n(result.mutable_data(), result.size(), T(0));
        T one(1), zero(0);
        int rc;
        {
            py::gil_scoped_release nogil;
            rc = htool_hmatrix_matmat(h, trans, &one, input.data(), mu, &zero, result.mutable_data());
        }
        check(rc);
        return result;
    }
    py::array_t<T, py::array::f_style> dense(bool user) const {
        py::array_t<T, py::array::f_style> out({(py::ssize_t)htool_hmatrix_nb_rows(h), (py::ssize_t)htool_hmatrix_nb_cols(h)});
        std::fill_n(out.mutable_data(), out.size(), T(0));
        check(htool_hmatrix_to_dense(h, out.mutable_data(), user ? 1 : 0));
        return out;
    }
};

template <typename T>
struct PyHMatrixTreeBuilder {
    htool_build_params p;
    std::shared_ptr<PyVirtualLowRankGenerator<T>> low_rank;
    std::shared_ptr<PyVirtualDenseBlocksGenerator<T>> dense_blocks;
    py::object low_rank_ref, dense_blocks_ref; // keep the Python subclasses (and their overrides) alive
    PyHMatrixTreeBuilder(double epsilon, double eta, char symmetry, char UPLO, int reqrank, std::shared_ptr<PyVirtualLowRankGenerator<T>> lr) : low_rank(lr) {
        htool_build_params_default(&p);
        p.epsilon = epsilon;
        p.eta = eta;
        p.symmetry = symmetry;
        p.uplo = UPLO;
        p.reqrank = reqrank;
    }
    htool_build_params resolved() const {
        htool_build_params q = p;
        if (low_rank) {
            q.compress = &PyVirtualLowRankGenerator<T>::trampoline;
            q.compress_ctx = low_rank.get();
            // allow_copy=False (virtual_low_rank_generator.hpp:39-42): the factors are borrowed from the Python arrays, which
            // the generator keeps alive until clear_data(); the library reads them once, when it ships the panels to HBM
            q.compress_borrows = low_rank->allow_copy ? 0 : 1;
        }
        if (dense_blocks) { q.dense_blocks = &PyVirtualDenseBlocksGenerator<T>::trampoline; q.dense_blocks_ctx = dense_blocks.get(); }
        return q;
    }
    // The reference builds on the Cluster object it is given.  Here builds take the ROOT of a tree plus a partition number; a
    // sub-cluster that is one of the tree's partitions (cluster.get_cluster_on_partition(p)) is mapped to that number, any
    // other sub-cluster is refused instead of silently building the whole-root operator.
    static int partition_of(const PyCluster &c, const char *which) {
        if (c.node == c.owner->root) return -1;
        for (int p = 0;; p++) {
            const htool_cluster *s = htool_cluster_on_partition(c.owner->root, p);
            if (!s) break;
            if (s == c.node) return p;
        }
        throw std::runtime_error(std::string("HMatrixTreeBuilder.build: the ") + which + " cluster is neither the root of its tree nor one of its partitions; pass the root "
                                 "cluster and target_partition_number (or use build_local for a (partition x partition) block)");
    }
    PyHMatrix<T> build(PyIGenerator<T> &generator, const PyCluster &target, const PyCluster &source, int target_partition_number, int partition_number_for_symmetry) const {
        const int tp = partition_of(target, "target"), sp = partition_of(source, "source");
        if (tp >= 0 && target_partition_number >= 0 && tp != target_partition_number) throw std::runtime_error("HMatrixTreeBuilder.build: target cluster and target_partition_number disagree");
        if (tp >= 0) target_partition_number = tp;
        if (sp >= 0) { // a source partition: the (target partition x source partition) block (as build_local)
            PyHMatrix<T> Hl;
            Hl.target = target;
            Hl.source = source;
            htool_build_params ql = resolved();
            check(htool_hmatrix_build_local(generator.get(), target.owner->root, source.owner->root, &ql, target_partition_number, sp, &Hl.h));
            return Hl;
        }
        PyHMatrix<T> H;
        H.target = target;
        H.source = source;
        htool_build_params q = resolved();
        htool_generator *g = generator.get();
        int rc;
        if (dynamic_cast<PyNativeGenerator<T> *>(&generator) && !low_rank && !dense_blocks) {
            py::gil_scoped_release nogil; // native generator: nothing calls back into Python during the build
            rc = htool_hmatrix_build(g, target.owner->root, source.owner->root, &q, target_partition_number, partition_number_for_symmetry, &H.h);
        } else { // callback generators / hooks run Python code on this thread: keep the GIL (CMakeLists.txt:97 rationale)
            rc = htool_hmatrix_build(g, target.owner->root, source.owner->root, &q, target_partition_number, partition_number_for_symmetry, &H.h);
        }
        check(rc);
        return H;
    }
};

// ---- RCCL communicator owned by the library (include/htool_mi355x.h: htool_comm_init_rccl) --------------------------
struct PyRcclCommunicator {
    htool_comm c;
    PyRcclCommunicator(py::bytes id, int rank, int size) {
        std::memset(&c, 0, sizeof(c));
        std::string raw = id;
        if (raw.size() != HTOOL_RCCL_UNIQUE_ID_BYTES) throw std::runtime_error("RcclCommunicator: the unique id must be the 128 bytes returned by Htool.rccl_unique_id() on one rank");
        int rc;
        {
            py::gil_scoped_release nogil; // collective: blocks until every rank has arrived
            rc = htool_comm_init_rccl(raw.data(), rank, size, &c);
        }
        check(rc);
    }
    PyRcclCommunicator(const PyRcclCommunicator &) = delete;
    ~PyRcclCommunicator() { htool_comm_destroy_rccl(&c); }
};

// ---- communicator adapter (stand-in for the mpi4py caster of src/htool/misc/wrapper_mpi.hpp:28-55) ----
struct PyComm {
    py::object obj;
    htool_comm c;
    static int allgatherv(void *ctx, const void *send, int64_t send_bytes, void *recv, const int64_t *recv_bytes, const int64_t *displs) {
        PyComm *self = static_cast<PyComm *>(ctx);
        try {
            int64_t total = 0;
            std::vector<int64_t> cnt(self->c.size), dsp(self->c.size);
            for (int p = 0; p < self->c.size; p++) { cnt[p] = recv_bytes[p]; dsp[p] = displs[p]; total = std::max(total, displs[p] + recv_bytes[p]); }
            py::array_t<unsigned char> s(std::array<py::ssize_t, 1>{(py::ssize_t)send_bytes}, (const unsigned char *)send, py::capsule(send, [](void *) {}));
            py::array_t<unsigned char> r(std::array<py::ssize_t, 1>{(py::ssize_t)total}, (unsigned char *)recv, py::capsule(recv, [](void *) {}));
            self->obj.attr("_htool_allgatherv")(s, r, cnt, dsp);
        } catch (py::error_already_set &e) {
            e.restore();
            return 1;
        }
        return 0;
    }
    explicit PyComm(py::object o) : obj(o) {
        std::memset(&c, 0, sizeof(c));
        c.rank = py::cast<int>(o.attr("Get_rank")());
        c.size = py::cast<int>(o.attr("Get_size")());
        // a communicator that carries a library-owned RCCL handle (Htool.RcclCommunicator, or the mpi4py stand-in after
        // use_rccl()): the exchange then runs inside the library, on device buffers
        py::object handle = py::hasattr(o, "_htool_comm_ptr") ? py::object(o.attr("_htool_comm_ptr")) : py::none();
        if (!handle.is_none()) {
            const htool_comm *lib = reinterpret_cast<const htool_comm *>(py::cast<std::uintptr_t>(handle));
            if (lib->rank != c.rank || lib->size != c.size) throw std::runtime_error("communicator object and its RCCL handle disagree on rank / size");
            c = *lib;
            return;
        }
        c.ctx = this;
        c.allgatherv = &allgatherv;
        if (c.size > 1 && !py::hasattr(o, "_htool_allgatherv")) throw std::runtime_error("communicator object must provide _htool_allgatherv (use the mpi4py shim shipped with this package)");
    }
};

template <typename T>
struct PyDistributedOperator {
    htool_distributed *d = nullptr;
    std::shared_ptr<PyComm> comm;
    PyHMatrix<T> *local = nullptr; // owned by the approximation builder that created this operator
    py::array_t<T, py::array::f_style> mul(const py::array_t<T, py::array::f_style> &input) const {
        int rows, cols;
        htool_distributed_shape(d, &rows, &cols);
        if (input.ndim() != 1) throw std::runtime_error("Wrong dimension for DistributedOperator-vector product");
        if (input.shape(0) != cols) throw std::runtime_error("Wrong size for DistributedOperator-vector product");
        py::array_t<T, py::array::f_style> result(std::array<py::ssize_t, 1>{rows});
        std::fill_n(result.mutable_data(), rows, T(0));
        check(htool_distributed_matvec(d, input.data(), result.mutable_data()));
        return result;
    }
    py::array_t<T, py::array::f_style> matmul(py::array_t<T, py::array::f_style> input) const {
        int rows, cols;
        htool_distributed_shape(d, &rows, &cols);
        if (input.ndim() != 2) throw std::runtime_error("Wrong dimension for HMatrix-matrix product");
        if (input.shape(0) != cols) throw std::runtime_error("Wrong size for HMatrix-matrix product");
        int mu = (int)input.shape(1);
        py::array_t<T, py::array::f_style> result(std::array<py::ssize_t, 2>{rows, mu});
        std::fill_n(result.mutable_data(), (size_t)rows * mu, T(0));
        check(htool_distributed_matmat(d, input.data(), mu, result.mutable_data()));
        return result;
    }
};

template <typename T>
struct PyDefaultApproximationBuilder {
    htool_distributed *d = nullptr;
    std::shared_ptr<PyComm> comm;
    PyCluster target, source;
    PyDistributedOperator<T> op;
    PyHMatrix<T> hmat, bdiag;
    py::object generator_ref, builder_ref; // the block-diagonal part is built on first use: keep what that build calls alive
    bool bdiag_asked = false;
    PyDefaultApproximationBuilder(py::object generator_obj, const PyCluster &t, const PyCluster &s, py::object builder_obj, py::object comm_obj)
        : comm(std::make_shared<PyComm>(comm_obj)), target(t), source(s), generator_ref(generator_obj), builder_ref(builder_obj) {
        PyIGenerator<T> &generator = generator_obj.cast<PyIGenerator<T> &>();
        const PyHMatrixTreeBuilder<T> &builder = builder_obj.cast<const PyHMatrixTreeBuilder<T> &>();
        htool_build_params q = builder.resolved();
        check(htool_distributed_create_default(generator.get(), t.owner->root, s.owner->root, &q, &comm->c, &d));
        op.d = d;
        op.comm = comm;
        op.local = &hmat;
        hmat.h = htool_distributed_hmatrix(d);
        hmat.owned = false;
        hmat.target = t;
        hmat.source = s;
        bdiag.owned = false;
        bdiag.target = t;
        bdiag.source = s;
    }
    PyHMatrix<T> *block_diagonal() {
        if (!bdiag_asked) {
            bdiag_asked = true;
            bdiag.h = htool_distributed_block_diagonal_hmatrix(d);
        }
        return bdiag.h ? &bdiag : nullptr;
    }
    ~PyDefaultApproximationBuilder() { htool_distributed_destroy(d); }
};

// ---- module ---------------------------------------------------------------------------------------
template <typename T>
static void declare_coefficient_classes(py::module &m, const std::string &prefix, const std::string &igenerator_name, const std::string &virtual_generator_name,
                                        const std::string &low_rank_generator_name, const std::string &native_name) {
    // generators (main.cpp:63,93)
    py::class_<PyIGenerator<T>>(m, igenerator_name.c_str());
    py::class_<PyVirtualGenerator<T>, PyIGenerator<T>, PyVirtualGeneratorTrampoline<T>>(m, virtual_generator_name.c_str())
        .def(py::init<>())
        .def(py::init<const py::array_t<int> &, const py::array_t<int> &>())
        .def("build_submatrix", &PyVirtualGenerator<T>::build_submatrix);
    py::class_<PyNativeGenerator<T>, PyIGenerator<T>>(m, native_name.c_str())
        .def(py::init<const std::string &, typename PyNativeGenerator<T>::coords_t, typename PyNativeGenerator<T>::coords_t, double>(), "kind"_a, "target_points"_a,
             "source_points"_a, "param"_a = 0.0);

    // LowRankMatrix (hmatrix/lrmat.hpp:15-17): introspection record
    struct LowRank { int m, n, r; };
    py::class_<LowRank>(m, (prefix + "LowRankMatrix").c_str())
        .def("nb_rows", [](const LowRank &l) { return l.m; })
        .def("nb_cols", [](const LowRank &l) { return l.n; })
        .def("rank", [](const LowRank &l) { return l.r; });

    // custom hooks (main.cpp:66-67,95-96)
    py::class_<PyVirtualLowRankGenerator<T>, std::shared_ptr<PyVirtualLowRankGenerator<T>>, PyVirtualLowRankGeneratorTrampoline<T>>(m, low_rank_generator_name.c_str())
        .def(py::init<bool>(), "allow_copy"_a = true)
        .def("build_low_rank_approximation", &PyVirtualLowRankGenerator<T>::build_low_rank_approximation)
        .def("set_U", &PyVirtualLowRankGenerator<T>::set_U)
        .def("set_V", &PyVirtualLowRankGenerator<T>::set_V)
        .def("clear_data", &PyVirtualLowRankGenerator<T>::clear_data);
    py::class_<PyVirtualDenseBlocksGenerator<T>, std::shared_ptr<PyVirtualDenseBlocksGenerator<T>>, PyVirtualDenseBlocksGeneratorTrampoline<T>>(
        m, (prefix + "VirtualDenseBlocksGenerator").c_str())
        .def(py::init<const PyCluster &, const PyCluster &>())
        .def("build_dense_blocks", &PyVirtualDenseBlocksGenerator<T>::build_dense_blocks);

    // HMatrix (hmatrix/hmatrix.hpp:27-138)
    typedef PyHMatrix<T> H;
    py::class_<H>(m, (prefix + "HMatrix").c_str())
        .def_property_readonly("shape", [](const H &s) { return std::pair<int, int>(htool_hmatrix_nb_rows(s.h), htool_hmatrix_nb_cols(s.h)); })
        .def("to_dense", [](const H &s) { return s.dense(false); })
        .def("to_dense_in_user_numbering", [](const H &s) { return s.dense(true); })
        .def("__deepcopy__", [](const H &s, py::dict) {
                H c;
                c.target = s.target;
                c.source = s.source;
                check(htool_hmatrix_clone(s.h, &c.h));
                return c;
            }, "memo"_a)
        .def("get_tree_parameters", [](const H &s) { return parse_info(s.h, 0); })
        .def("get_local_information", [](const H &s) { return parse_info(s.h, 1); })
        .def("get_distributed_information", [](const H &s, py::object) { return parse_info(s.h, 1); })
        .def("get_target_cluster", [](const H &s) { return PyCluster{s.target.owner, htool_hmatrix_target_cluster(s.h)}; })
        .def("get_source_cluster", [](const H &s) { return PyCluster{s.source.owner, htool_hmatrix_source_cluster(s.h)}; })
        .def("lu_factorization", [](H &s) { py::gil_scoped_release nogil; check(htool_hmatrix_lu_factorization(s.h)); })
        // extensions of the device path of the dense fallback (include/htool_mi355x.h): LU of (H + shift I), solves on device
        // right-hand sides in the operator's cluster numbering, the dense expansion on the device
        .def("lu_factorization_shifted", [](H &s, double shift) { py::gil_scoped_release nogil; check(htool_hmatrix_lu_factorization_shifted(s.h, shift)); }, "shift"_a)
        .def("factor_solve_device", [](const H &s, int kind, char trans, std::uintptr_t b_dev, long long ldb, int mu, std::uintptr_t stream) {
                check(htool_hmatrix_factor_solve_device(s.h, kind, trans, (void *)b_dev, ldb, mu, (void *)stream));
            }, "kind"_a, "trans"_a, "b_ptr"_a, "ldb"_a, "mu"_a, "stream"_a = 0)
        .def("to_dense_device", [](const H &s, std::uintptr_t out_dev, long long ld, std::uintptr_t stream) {
                py::gil_scoped_release nogil;
                check(htool_hmatrix_to_dense_device(s.h, (void *)out_dev, ld, (void *)stream));
            }, "out_ptr"_a, "ld"_a, "stream"_a = 0)
        .def("cholesky_factorization", [](H &s, char UPLO) { py::gil_scoped_release nogil; check(htool_hmatrix_cholesky_factorization(s.h, UPLO)); })
        .def("factorization_info", [](const H &s) {
                int64_t v[17];
                double sec[4];
                check(htool_hmatrix_factorization_info(s.h, v, sec));
                py::dict d;
                static const char *kinds[] = {"none", "dense host", "dense device", "hierarchical"};
                d["kind"] = kinds[v[0] < 0 || v[0] > 3 ? 0 : v[0]];
                if (v[0] == 3) {
                    static const char *names[] = {"unknowns", "leaves", "tasks", "launches", "windows", "factor_bytes", "peak_bytes", "truncations_at_capacity", "truncations", "appended_columns",
                                                  "dense_product_columns", "solve_tasks", "solve_launches", "rank_weight", "rows_plus_columns", "eps_e12"};
                    for (int i = 0; i < 16; i++) d[names[i]] = v[1 + i];
                    d["plan_s"] = sec[0]; d["unpack_s"] = sec[1]; d["factor_s"] = sec[2]; d["total_s"] = sec[3];
                }
                return d;
            }, "What the last lu_factorization / cholesky_factorization left behind: kind = 'hierarchical' (device H-LU, the default), 'dense device', 'dense host' or 'none', with the "
               "statistics of a hierarchical factorisation")
        .def("lu_solve", [](const H &s, char trans, const py::array_t<T, py::array::f_style> &input) {
                if (input.ndim() != 1 && input.ndim() != 2) throw std::runtime_error("Wrong dimension for HMatrix-LU input");
                py::array_t<T, py::array::f_style> result = input.ndim() == 1 ? py::array_t<T, py::array::f_style>(input.shape(0))
                                                                             : py::array_t<T, py::array::f_style>({input.shape(0), input.shape(1)});
                std::copy_n(input.data(), input.size(), result.mutable_data());
                if (input.shape(0) != htool_hmatrix_nb_rows(s.h)) throw std::runtime_error("Wrong size for HMatrix-LU input");
                check(htool_hmatrix_factor_solve(s.h, 1, trans, result.mutable_data(), input.ndim() == 1 ? 1 : (int)input.shape(1)));
                return result;
            })
        .def("cholesky_solve", [](const H &s, char UPLO, const py::array_t<T, py::array::f_style> &input) {
                if (input.ndim() != 1 && input.ndim() != 2) throw std::runtime_error("Wrong dimension for HMatrix-Cholesky input");
                py::array_t<T, py::array::f_style> result = input.ndim() == 1 ? py::array_t<T, py::array::f_style>(input.shape(0))
                                                                             : py::array_t<T, py::array::f_style>({input.shape(0), input.shape(1)});
                std::copy_n(input.data(), input.size(), result.mutable_data());
                if (input.shape(0) != htool_hmatrix_nb_rows(s.h)) throw std::runtime_error("Wrong size for HMatrix-Cholesky input");
                (void)UPLO;
                check(htool_hmatrix_factor_solve(s.h, 2, 'N', result.mutable_data(), input.ndim() == 1 ? 1 : (int)input.shape(1)));
                return result;
            })
        .def("__mul__", &H::mul, "in"_a)
        .def("__matmul__", &H::matmul, "in"_a)
        // extensions used by tests / bench: flattened leaf table, leaf panels, statistics
        .def("leaves", [](const H &s) {
                int64_t n = htool_hmatrix_leaf_count(s.h);
                py::array_t<int> out({(py::ssize_t)n, (py::ssize_t)5});
                htool_hmatrix_leaves(s.h, out.mutable_data());
                return out;
            })
        .def("leaf_panels", [](const H &s, int64_t i) -> py::object {
                int64_t n = htool_hmatrix_leaf_count(s.h);
                if (i < 0 || i >= n) throw std::runtime_error("leaf index out of range");
                std::vector<int> all((size_t)n * 5);
                htool_hmatrix_leaves(s.h, all.data());
                int mm = all[5 * i + 1], nn = all[5 * i + 3], r = all[5 * i + 4];
                if (r < 0) {
                    py::array_t<T, py::array::f_style> A({(py::ssize_t)mm, (py::ssize_t)nn});
                    check(htool_hmatrix_leaf_panels(s.h, i, A.mutable_data(), nullptr));
                    return py::make_tuple(A, py::none());
                }
                py::array_t<T, py::array::f_style> U({(py::ssize_t)mm, (py::ssize_t)r}), V({(py::ssize_t)r, (py::ssize_t)nn});
                if (r > 0) check(htool_hmatrix_leaf_panels(s.h, i, U.mutable_data(), V.mutable_data()));
                return py::make_tuple(U, V);
            })
        .def("leaf_panels_bulk", [](const H &s, py::array_t<int64_t, py::array::c_style | py::array::forcecast> ids) {
                const int64_t n = ids.size();
                py::array_t<int64_t> offs({(py::ssize_t)n, (py::ssize_t)2});
                int64_t elems = 0;
                check(htool_hmatrix_leaf_panels_bulk(s.h, n, ids.data(), offs.mutable_data(), nullptr, &elems));
                py::array_t<T> out((py::ssize_t)elems);
                check(htool_hmatrix_leaf_panels_bulk(s.h, n, ids.data(), offs.mutable_data(), out.mutable_data(), &elems));
                return py::make_tuple(offs, out);
            }, "leaf_ids"_a)
        .def("stats", [](const H &s) {
                int64_t st[8];
                htool_hmatrix_stats(s.h, st);
                py::dict d;
                d["dense_elements"] = st[0]; d["low_rank_elements"] = st[1]; d["n_dense"] = st[2]; d["n_low_rank"] = st[3];
                d["sum_rank"] = st[4]; d["hbm_bytes"] = st[5]; d["build_seconds"] = st[6] * 1e-6; d["max_rank"] = st[7];
                return d;
            })
        .def("is_one_triangle", [](const H &s) { return htool_hmatrix_is_one_triangle(s.h) != 0; })
        .def("set_phase_timing", [](const H &s, bool on) { check(htool_hmatrix_set_phase_timing(s.h, on ? 1 : 0)); }, "on"_a = true)
        .def("last_product_us", [](const H &s) { return htool_hmatrix_last_product_us(s.h); })
        .def("phase_times_us", [](const H &s) {
                double t[4];
                int n = htool_hmatrix_phase_times(s.h, t);
                return py::make_tuple(n, std::vector<double>(t, t + 4));
            })
        .def("matvec_device", [](const H &s, std::uintptr_t x_dev, std::uintptr_t y_dev, int numbering, std::uintptr_t stream) {
                check(htool_hmatrix_matvec_device(s.h, (const void *)x_dev, (void *)y_dev, numbering, (void *)stream));
            }, "x_ptr"_a, "y_ptr"_a, "numbering"_a = 0, "stream"_a = 0)
        .def("matmat_device", [](const H &s, std::uintptr_t x_dev, long long ldx, std::uintptr_t y_dev, long long ldy, int mu, int numbering, std::uintptr_t stream) {
                check(htool_hmatrix_matmat_device(s.h, (const void *)x_dev, ldx, (void *)y_dev, ldy, mu, numbering, (void *)stream));
            }, "x_ptr"_a, "ldx"_a, "y_ptr"_a, "ldy"_a, "mu"_a, "numbering"_a = 0, "stream"_a = 0)
        .def("matmat_device_trans", [](const H &s, char trans, std::uintptr_t x_dev, long long ldx, std::uintptr_t y_dev, long long ldy, int mu, int numbering, std::uintptr_t stream) {
                check(htool_hmatrix_matmat_device_trans(s.h, trans, (const void *)x_dev, ldx, (void *)y_dev, ldy, mu, numbering, (void *)stream));
            }, "trans"_a, "x_ptr"_a, "ldx"_a, "y_ptr"_a, "ldy"_a, "mu"_a, "numbering"_a = 0, "stream"_a = 0)
        .def("transposed_mul", &H::transposed_mul, "x"_a, "trans"_a = 'T',
             "extension: H^T x (trans='T') or H^H x ('C') for a vector or the columns of a matrix; x has one entry per row of H")
        .def_property_readonly("_handle", [](const H &s) { return (std::uintptr_t)s.h; });

    // checkpoint / resume: rebuild an H-matrix from leaves and panels saved by htool_python_amd/io.py
    m.def(("_" + std::string(std::is_same<T, double>::value ? "" : "complex_") + "hmatrix_from_leaves").c_str(),
          [](const PyCluster &target, const PyCluster &source, double epsilon, double eta, char symmetry, char UPLO, bool one_triangle, int target_partition_number,
             py::array_t<int, py::array::c_style | py::array::forcecast> leaves, py::array_t<int64_t, py::array::c_style | py::array::forcecast> offsets,
             py::array_t<T, py::array::c_style | py::array::forcecast> data) {
              if (leaves.ndim() != 2 || leaves.shape(1) != 5 || offsets.ndim() != 2 || offsets.shape(1) != 2 || offsets.shape(0) != leaves.shape(0) || data.ndim() != 1)
                  throw std::runtime_error("hmatrix_from_leaves: leaves must be (n, 5), offsets (n, 2), data one-dimensional");
              htool_build_params p;
              htool_build_params_default(&p);
              p.epsilon = epsilon; p.eta = eta; p.symmetry = symmetry; p.uplo = UPLO; p.store_one_triangle = one_triangle ? 1 : 0;
              H out;
              out.target = target;
              out.source = source;
              check(htool_hmatrix_build_from_leaves(target.owner->root, source.owner->root, &p, std::is_same<T, double>::value ? 0 : 1, target_partition_number,
                                                    (int64_t)leaves.shape(0), leaves.data(), offsets.data(), data.data(), (int64_t)data.size(), &out.h));
              return out;
          }, "target_cluster"_a, "source_cluster"_a, "epsilon"_a, "eta"_a, "symmetry"_a, "UPLO"_a, "one_triangle"_a, "target_partition_number"_a, "leaves"_a,
          "offsets"_a, "data"_a);

    // recompression / openmp_recompression (hmatrix/hmatrix.hpp:96-99): SVD recompression on the device; the variant
    // taking a Python callable per low-rank matrix has no device counterpart and falls back to the built-in rule
    auto recompress = [](H &s, double epsilon = -1.0) { // epsilon <= 0: the tolerance the H-matrix was built with
        int64_t n = 0;
        check(htool_hmatrix_recompress(s.h, epsilon, &n));
        return n;
    };
    m.def("recompression", [recompress](H &s) { return recompress(s); });
    m.def("recompression", [recompress](H &s, double epsilon) { return recompress(s, epsilon); }, "hmatrix"_a, "epsilon"_a); // extension: explicit tolerance
    // recompression(hmatrix, fn) (hmatrix.hpp:96): the reference hands every low-rank leaf to fn as a LowRankMatrix INSTEAD of
    // applying its built-in rule; through the binding fn sees nb_rows / nb_cols / rank (lrmat.hpp:15-17) and nothing it could
    // change, so the call is a visit of the low-rank leaves: reproduced as such (host callback per leaf, nothing recompressed).
    auto visit_low_rank = [](H &s, py::function fn) {
        const int64_t n = htool_hmatrix_leaf_count(s.h);
        std::vector<int> all((size_t)n * 5);
        htool_hmatrix_leaves(s.h, all.data());
        int64_t visited = 0;
        for (int64_t i = 0; i < n; i++)
            if (all[5 * i + 4] >= 0) { fn(LowRank{all[5 * i + 1], all[5 * i + 3], all[5 * i + 4]}); visited++; }
        return visited;
    };
    m.def("recompression", [visit_low_rank](H &s, py::function fn) { return visit_low_rank(s, fn); }, "hmatrix"_a, "recompression_function"_a);
    m.def("openmp_recompression", [recompress](H &s) { return recompress(s); });
    m.def("openmp_recompression", [recompress](H &s, double epsilon) { return recompress(s, epsilon); }, "hmatrix"_a, "epsilon"_a);
    m.def("openmp_recompression", [visit_low_rank](H &s, py::function fn) { return visit_low_rank(s, fn); }, "hmatrix"_a, "recompression_function"_a);

    // HMatrixTreeBuilder (hmatrix/hmatrix_tree_builder.hpp:10-44)
    typedef PyHMatrixTreeBuilder<T> B;
    py::class_<B>(m, (prefix + "HMatrixTreeBuilder").c_str())
        .def(py::init([](double epsilon, double eta, char symmetry, char UPLO, int reqrank, py::object low_rank_strategy) {
                 std::shared_ptr<PyVirtualLowRankGenerator<T>> lr;
                 if (!low_rank_strategy.is_none()) lr = low_rank_strategy.cast<std::shared_ptr<PyVirtualLowRankGenerator<T>>>();
                 auto b = std::make_unique<B>(epsilon, eta, symmetry, UPLO, reqrank, lr);
                 if (lr) b->low_rank_ref = low_rank_strategy;
                 return b;
             }), "epsilon"_a, "eta"_a, "symmetry"_a, "UPLO"_a, py::kw_only(), "reqrank"_a = -1, "low_rank_strategy"_a = py::none())
        .def("build", &B::build, "generator"_a, "target_cluster"_a, "source_cluster"_a, "target_partition_number"_a = -1, "partition_number_for_symmetry"_a = -1)
        .def("build_local", [](const B &b, PyIGenerator<T> &generator, const PyCluster &target, const PyCluster &source, int target_partition_number, int source_partition_number) {
                PyHMatrix<T> H;
                H.target = target;
                H.source = source;
                htool_build_params q = b.resolved();
                check(htool_hmatrix_build_local(generator.get(), target.owner->root, source.owner->root, &q, target_partition_number, source_partition_number, &H.h));
                return H;
            }, "generator"_a, "target_cluster"_a, "source_cluster"_a, "target_partition_number"_a, "source_partition_number"_a)
        .def("set_minimal_source_depth", [](B &b, int d) { b.p.minimal_source_depth = d; })
        .def("set_minimal_target_depth", [](B &b, int d) { b.p.minimal_target_depth = d; })
        .def("set_low_rank_generator", [](B &b, py::object g) { b.low_rank = g.cast<std::shared_ptr<PyVirtualLowRankGenerator<T>>>(); b.low_rank_ref = g; })
        .def("set_dense_blocks_generator", [](B &b, py::object g) { b.dense_blocks = g.cast<std::shared_ptr<PyVirtualDenseBlocksGenerator<T>>>(); b.dense_blocks_ref = g; })
        .def("set_block_tree_consistency", [](B &b, bool c) { b.p.block_tree_consistency = c ? 1 : 0; })
        // extension: False stores both triangles of a symmetric operator (default: the UPLO triangle only, as the reference does)
        .def("set_aca_confirmation_steps", [](B &b, int steps) {
                if (steps < 0 || steps > 8) throw std::runtime_error("set_aca_confirmation_steps: between 0 and 8");
                b.p.aca_confirm_steps = steps;
            }, "steps"_a,
             "Extension (htool_build_params.aca_confirm_steps).  0 (default): the reference's stopping rule of the partially pivoted ACA.  c > 0: a leaf is "
             "accepted only after c further pivot steps have passed the stopping test too; the confirming terms are dropped, so leaves on which the test "
             "was right keep exactly their factors -- a safeguard for nearly collinear point clouds, where partial pivoting can stop far too early.")
        .def("set_transposed_products", [](B &b, bool at_build) { b.p.transposed_products = at_build ? 1 : 0; }, "at_build"_a = true,
             "Extension: lay the index tables of the transposed product (H^T x, H^H x) out during the build instead of at the first such product")
        .def("set_symmetric_storage", [](B &b, bool one_triangle) { b.p.store_one_triangle = one_triangle ? 1 : 0; }, "one_triangle"_a,
             "True (default): symmetry 'S'/'H' keeps the UPLO triangle only, as the reference does; every product uses each stored leaf "
             "twice in one fused sweep (half the memory, about 1.5x faster per vector above ~20 000 unknowns).  False: both triangles "
             "are stored; preferable when products mostly come with many right-hand sides (H @ X sweeps 8 columns per pass then, "
             "4 columns per fused pass in one-triangle storage: about 1.5x faster for 8 columns, twice the memory) or for very small operators.");

    // DistributedOperator + DefaultApproximationBuilder (distributed_operator/*.hpp)
    typedef PyDistributedOperator<T> Op;
    py::class_<Op>(m, (prefix + "DistributedOperator").c_str())
        .def_property_readonly("shape", [](const Op &o) {
                int r, c;
                htool_distributed_shape(o.d, &r, &c);
                return std::pair<int, int>(r, c);
            })
        .def("__mul__", &Op::mul, "in"_a)
        .def("__matmul__", &Op::matmul, py::arg("input").noconvert(true))
        // extensions used by the GPU-resident Krylov loop (htool_python_amd/solver.py)
        .def_property_readonly("local_hmatrix", [](Op &o) { return o.local; }, py::return_value_policy::reference_internal)
        // GPU-resident product: this rank's slice of x in, this rank's rows of y out (device pointers, cluster numbering);
        // the exchange is an RCCL all-gather inside the library (htool_distributed_matvec_device)
        .def("matvec_device", [](Op &o, std::uintptr_t x_local, std::uintptr_t y_local, std::uintptr_t stream) {
                check(htool_distributed_matvec_device(o.d, (const void *)x_local, (void *)y_local, (void *)stream));
            }, "x_local_ptr"_a, "y_local_ptr"_a, "stream"_a = 0)
        .def("matmat_device", [](Op &o, std::uintptr_t x_local, long long ldx, std::uintptr_t y_local, long long ldy, int mu, std::uintptr_t stream) {
                check(htool_distributed_matmat_device(o.d, (const void *)x_local, ldx, (void *)y_local, ldy, mu, (void *)stream));
            }, "x_local_ptr"_a, "ldx"_a, "y_local_ptr"_a, "ldy"_a, "mu"_a, "stream"_a = 0)
        .def("matmat_device_trans", [](Op &o, char trans, std::uintptr_t x_local, long long ldx, std::uintptr_t y_local, long long ldy, int mu, std::uintptr_t stream) {
                check(htool_distributed_matmat_device_trans(o.d, trans, (const void *)x_local, ldx, (void *)y_local, ldy, mu, (void *)stream));
            }, "trans"_a, "x_local_ptr"_a, "ldx"_a, "y_local_ptr"_a, "ldy"_a, "mu"_a, "stream"_a = 0)
        .def_property_readonly("has_rccl", [](Op &o) { return o.comm->c.rccl != nullptr; })
        // which exchange matvec_device / matmat_device run (htool_distributed_exchange_kind): 0 none, 1 zero-copy, 2 padded + compaction,
        // 3 / 4 the same staged through the host all-gather of the communicator object, -1 not possible
        .def("exchange_kind", [](Op &o, int mu) { return htool_distributed_exchange_kind(o.d, mu); }, "mu"_a = 1)
        .def_property_readonly("comm", [](Op &o) { return o.comm->obj; })
        .def("partition", [](Op &o) {
                std::vector<std::pair<int, int>> out;
                for (int p = 0; p < o.comm->c.size; p++) {
                    int off = 0, sz = 0;
                    check(htool_distributed_partition(o.d, p, &off, &sz));
                    out.emplace_back(off, sz);
                }
                return out;
            });
    typedef PyDefaultApproximationBuilder<T> DA;
    py::class_<DA>(m, (prefix + "DefaultApproximationBuilder").c_str())
        .def(py::init<py::object, const PyCluster &, const PyCluster &, py::object, py::object>())
        .def_property_readonly("distributed_operator", [](DA &s) { return &s.op; }, py::return_value_policy::reference_internal)
        .def_property_readonly("hmatrix", [](DA &s) { return &s.hmat; }, py::return_value_policy::reference_internal)
        .def_property_readonly("block_diagonal_hmatrix", [](DA &s) -> PyHMatrix<T> * { return s.block_diagonal(); }, py::return_value_policy::reference_internal);
}

PYBIND11_MODULE(Htool, m) {
    m.doc() = "MI355X-native H-matrix engine behind the Htool Python API";
    htool_set_log_sink(&python_log_sink);
    m.def("test_logger", &htool_test_logger);
    m.def("device_count", &htool_device_count);
    m.def("device_name", []() { return std::string(htool_device_name()); });
    m.def("set_device", [](int d) { check(htool_set_device(d)); });
    m.def("last_warm_up_seconds", &htool_last_warm_up_seconds, "Seconds the library warm-up of the most recent set_device took (code objects, streams; 0 when the device was warm already)");
    m.def("set_num_threads", &htool_set_num_threads);
    m.def("rccl_unique_id", []() {
        char id[HTOOL_RCCL_UNIQUE_ID_BYTES];
        check(htool_rccl_get_unique_id(id));
        return py::bytes(id, sizeof(id));
    }, "128-byte id for RcclCommunicator: call on ONE rank and broadcast the bytes to the others");
    py::class_<PyRcclCommunicator>(m, "RcclCommunicator", "RCCL communicator owned by the library (one process per GPU); accepted wherever the reference takes mpi4py.MPI.COMM_WORLD")
        .def(py::init<py::bytes, int, int>(), "unique_id"_a, "rank"_a, "size"_a)
        .def("Get_rank", [](const PyRcclCommunicator &s) { return s.c.rank; })
        .def("Get_size", [](const PyRcclCommunicator &s) { return s.c.size; })
        .def_property_readonly("rank", [](const PyRcclCommunicator &s) { return s.c.rank; })
        .def_property_readonly("size", [](const PyRcclCommunicator &s) { return s.c.size; })
        .def_property_readonly("_htool_comm_ptr", [](const PyRcclCommunicator &s) { return (std::uintptr_t)&s.c; });

    py::class_<PyCluster>(m, "Cluster")
        .def("get_size", [](const PyCluster &c) { return htool_cluster_size(c.node); })
        .def("get_offset", [](const PyCluster &c) { return htool_cluster_offset(c.node); })
        .def("get_maximal_leaf_size", [](const PyCluster &c) { return htool_cluster_maximal_leaf_size(c.node); })
        .def("get_permutation", [](py::object self) {
                const PyCluster &c = self.cast<const PyCluster &>();
                int n = 0;
                const int *p = htool_cluster_permutation(c.node, &n);
                return py::array_t<int>(std::array<py::ssize_t, 1>{n}, p, self); // view; keeps the cluster alive
            })
        .def("get_cluster_on_partition", [](const PyCluster &c, int p) {
                const htool_cluster *s = htool_cluster_on_partition(c.node, p);
                if (!s) throw std::runtime_error(htool_last_error());
                return PyCluster{c.owner, s};
            })
        .def("_nodes", [](const PyCluster &c) {
                int n = htool_cluster_node_count(c.node);
                py::array_t<int> ints({(py::ssize_t)n, (py::ssize_t)7});
                py::array_t<double> dbl({(py::ssize_t)n, (py::ssize_t)4});
                htool_cluster_nodes(c.node, ints.mutable_data(), dbl.mutable_data());
                return py::make_tuple(ints, dbl);
            })
        .def("_node_id", [](const PyCluster &c) { return htool_cluster_node_id(c.node); })
        .def("_dimension", [](const PyCluster &c) { return htool_cluster_dimension(c.node); })
        .def("_number_of_children", [](const PyCluster &c) { return htool_cluster_number_of_children(c.node); });

    m.def("_cluster_from_tables", [](int dim, int maximal_leaf_size, int number_of_children, py::array_t<int, py::array::c_style | py::array::forcecast> permutation,
                                     py::array_t<int, py::array::c_style | py::array::forcecast> ints7, py::array_t<double, py::array::c_style | py::array::forcecast> doubles4) {
        if (permutation.ndim() != 1 || ints7.ndim() != 2 || ints7.shape(1) != 7 || doubles4.ndim() != 2 || doubles4.shape(1) != 4 || doubles4.shape(0) != ints7.shape(0))
            throw std::runtime_error("cluster tables: permutation (n), node ints (k, 7), node doubles (k, 4) expected");
        auto owner = std::make_shared<ClusterRoot>();
        check(htool_cluster_create_from_tables((int)permutation.shape(0), dim, maximal_leaf_size, number_of_children, permutation.data(), (int)ints7.shape(0), ints7.data(), doubles4.data(),
                                               &owner->root));
        return PyCluster{owner, owner->root};
    }, "dim"_a, "maximal_leaf_size"_a, "number_of_children"_a, "permutation"_a, "node_ints"_a, "node_doubles"_a);

    py::class_<PyPartitioning, std::shared_ptr<PyPartitioning>>(m, "VirtualPartitioning");
    py::class_<PyPartitioningT<HTOOL_PCA_REGULAR>, std::shared_ptr<PyPartitioningT<HTOOL_PCA_REGULAR>>, PyPartitioning>(m, "PCARegular").def(py::init<>());
    py::class_<PyPartitioningT<HTOOL_PCA_GEOMETRIC>, std::shared_ptr<PyPartitioningT<HTOOL_PCA_GEOMETRIC>>, PyPartitioning>(m, "PCAGeometric").def(py::init<>());
    py::class_<PyPartitioningT<HTOOL_BBOX_REGULAR>, std::shared_ptr<PyPartitioningT<HTOOL_BBOX_REGULAR>>, PyPartitioning>(m, "BoundingBoxRegular").def(py::init<>());
    py::class_<PyPartitioningT<HTOOL_BBOX_GEOMETRIC>, std::shared_ptr<PyPartitioningT<HTOOL_BBOX_GEOMETRIC>>, PyPartitioning>(m, "BoundingBoxGeometric").def(py::init<>());

    typedef PyClusterTreeBuilder CB;
    py::class_<CB>(m, "ClusterTreeBuilder")
        .def(py::init<>())
        .def("create_cluster_tree", [](CB &self, CB::coords_t coordinates, int number_of_children, std::optional<int> size_of_partition, std::optional<CB::coords_t> radii,
                                       std::optional<CB::coords_t> weights) {
                return self.create(coordinates, number_of_children, size_of_partition ? *size_of_partition : number_of_children, nullptr, false, radii, weights);
            // the reference binding makes size_of_partition keyword-only (cluster_tree_builder.hpp:28-31) while its own
            // example passes it positionally (example/use_ddm_solver.py:30-32): both spellings are accepted here
            }, "coordinates"_a, "number_of_children"_a, "size_of_partition"_a = py::none(), py::kw_only(), "radii"_a = py::none(), "weights"_a = py::none())
        .def("create_cluster_tree_from_global_partition", [](CB &self, CB::coords_t coordinates, int number_of_children, int size_of_partition, CB::part_t partition,
                                                             std::optional<CB::coords_t> radii, std::optional<CB::coords_t> weights) {
                if (partition.ndim() != 1 || partition.shape(0) != coordinates.shape(1)) throw std::runtime_error("Wrong format for partition");
                return self.create(coordinates, number_of_children, size_of_partition, partition.data(), false, radii, weights);
            }, "coordinates"_a, "number_of_children"_a, "size_of_partition"_a, "partition"_a, py::kw_only(), "radii"_a = py::none(), "weights"_a = py::none())
        .def("create_cluster_tree_from_local_partition", [](CB &self, CB::coords_t coordinates, int number_of_children, int size_of_partition, CB::part_t partition,
                                                            std::optional<CB::coords_t> radii, std::optional<CB::coords_t> weights) {
                if (partition.ndim() != 2 || partition.shape(0) != 2 || partition.shape(1) != size_of_partition) throw std::runtime_error("Wrong format for partition");
                return self.create(coordinates, number_of_children, size_of_partition, partition.data(), true, radii, weights);
            }, "coordinates"_a, "number_of_children"_a, "size_of_partition"_a, "partition"_a, py::kw_only(), "radii"_a = py::none(), "weights"_a = py::none())
        .def("set_maximal_leaf_size", [](CB &self, int s) { self.max_leaf = s; })
        .def("set_partitioning_strategy", [](CB &self, std::shared_ptr<PyPartitioning> p) { self.strategy = p->strategy; });

    // host-only introspection (no GPU needed): block-tree work queues and tile partition
    m.def("block_tree_queues", [](const PyCluster &t, const PyCluster &s, double eta, int min_target_depth, int min_source_depth, int target_partition_number,
                                  char symmetry, char UPLO, bool one_triangle) {
            htool_build_params p;
            htool_build_params_default(&p);
            p.eta = eta; p.minimal_target_depth = min_target_depth; p.minimal_source_depth = min_source_depth;
            p.symmetry = symmetry; p.uplo = UPLO; p.store_one_triangle = one_triangle ? 1 : 0;
            int64_t na = 0, nd = 0;
            check(htool_block_tree_queues(t.owner->root, s.owner->root, &p, target_partition_number, &na, &nd, nullptr, nullptr));
            py::array_t<int> a({(py::ssize_t)na, (py::ssize_t)4}), d({(py::ssize_t)nd, (py::ssize_t)4});
            check(htool_block_tree_queues(t.owner->root, s.owner->root, &p, target_partition_number, &na, &nd, a.mutable_data(), d.mutable_data()));
            return py::make_tuple(a, d);
        }, "target_cluster"_a, "source_cluster"_a, "eta"_a, "min_target_depth"_a = 0, "min_source_depth"_a = 0, "target_partition_number"_a = -1,
           "symmetry"_a = 'N', "UPLO"_a = 'N', "one_triangle"_a = true);
    m.def("krylov_finish_step", [](std::uintptr_t W, long long ldw, int n, int mu, bool is_complex, std::uintptr_t h1, std::uintptr_t t2, int j, std::uintptr_t mask,
                                   std::uintptr_t coef, bool scale, std::uintptr_t V, long long ld_basis, long long ld_rhs, std::uintptr_t stream) {
            check(htool_krylov_finish_step((void *)W, ldw, n, mu, is_complex ? 1 : 0, (const void *)h1, (const void *)t2, j, (const double *)mask, (void *)coef, scale ? 1 : 0,
                                           (const void *)V, ld_basis, ld_rhs, (void *)stream));
        }, "W_ptr"_a, "ldw"_a, "n"_a, "mu"_a, "is_complex"_a, "h1_ptr"_a, "t2_ptr"_a, "j"_a, "mask_ptr"_a, "coef_ptr"_a, "scale"_a, "V_ptr"_a, "ld_basis"_a, "ld_rhs"_a, "stream"_a,
        "Tail of a GMRES step in one launch (htool_krylov_finish_step): second projection taken out, norm, scaling of w, coefficient row for the host");
    m.def("release_workspace", []() { return htool_release_workspace(); }, "Free the cached temporary device buffers of builds / recompressions; returns the bytes released");
    m.def("cluster_tiles", [](const PyCluster &c, int partition_number, int tile_max) {
            int n = htool_cluster_tiles(c.owner->root, partition_number, tile_max, nullptr, 0);
            py::array_t<int> out({(py::ssize_t)n, (py::ssize_t)2});
            htool_cluster_tiles(c.owner->root, partition_number, tile_max, out.mutable_data(), n);
            return out;
        }, "cluster"_a, "partition_number"_a = -1, "tile_max"_a = 128);

    // the plan of the hierarchical LU (host only; tests hand its task lists to the CPU checker under oracle/)
    struct PyHluPlan {
        htool_hlu_plan *p = nullptr;
        std::shared_ptr<ClusterRoot> owner;
        ~PyHluPlan() { htool_hlu_plan_free(p); }
    };
    py::class_<PyHluPlan, std::shared_ptr<PyHluPlan>>(m, "HLUPlan")
        .def(py::init([](const PyCluster &c, py::array_t<int32_t, py::array::c_style | py::array::forcecast> rects5, double epsilon, int cap_min, int cap_max,
                         long long window_scratch_elems, long long window_tasks, double cap_factor, bool symmetric, int super_rows, int solve_slots) {
                 if (rects5.ndim() != 2 || rects5.shape(1) != 5) throw std::runtime_error("HLUPlan: rects must be (n_leaves, 5) int32: t_off, m, s_off, n, rank");
                 auto pl = std::make_shared<PyHluPlan>();
                 pl->owner = c.owner;
                 {
                     py::gil_scoped_release nogil;
                     check(htool_hlu_plan_create(c.node, rects5.shape(0), rects5.data(), epsilon, cap_min, cap_max, cap_factor, window_scratch_elems, window_tasks, symmetric ? 1 : 0, super_rows, solve_slots, &pl->p));
                 }
                 return pl;
             }), "cluster"_a, "rects"_a, "epsilon"_a = 1e-3, "cap_min"_a = 0, "cap_max"_a = 0, "window_scratch_elems"_a = 0, "window_tasks"_a = 0, "cap_factor"_a = 0.0, "symmetric"_a = false, "super_rows"_a = -1, "solve_slots"_a = -1)
        .def("info", [](const PyHluPlan &s) {
                py::array_t<int64_t> out(23);
                check(htool_hlu_plan_info(s.p, out.mutable_data(), 23));
                return out;
            })
        .def("program", [](const PyHluPlan &s, int which) { // copies: (tasks as bytes (n, 96), buckets as int64 (n, 5), runs, scratch elements)
                const void *tasks, *buckets;
                const int64_t *seg, *aux;
                int64_t nt, nb, ns, scratch, na;
                check(htool_hlu_plan_program(s.p, which, &tasks, &nt, &buckets, &nb, &seg, &ns, &scratch, &aux, &na));
                py::array_t<int64_t> ax((py::ssize_t)na);
                if (na) std::memcpy(ax.mutable_data(), aux, (size_t)na * 8);
                py::array_t<uint8_t> t({(py::ssize_t)nt, (py::ssize_t)96});
                std::memcpy(t.mutable_data(), tasks, (size_t)nt * 96);
                py::array_t<int64_t> b({(py::ssize_t)nb, (py::ssize_t)5});
                std::memcpy(b.mutable_data(), buckets, (size_t)nb * 40);
                py::array_t<int64_t> g((py::ssize_t)ns);
                std::memcpy(g.mutable_data(), seg, (size_t)ns * 8);
                return py::make_tuple(t, b, g, scratch, ax);
            }, "which"_a)
        .def("debug_execute", [](const PyHluPlan &s, int first, int last, py::array_t<double> factor, py::array_t<double> diag, py::array_t<int32_t> rank,
                                 py::array_t<double> norm0, py::array_t<double> norm2, py::array_t<int64_t> counters, py::object rhs, long long ld_rhs, int nrhs, py::object scratch) {
                double *r = nullptr, *sc = nullptr;
                if (!rhs.is_none()) r = rhs.cast<py::array_t<double>>().mutable_data();
                if (!scratch.is_none()) sc = scratch.cast<py::array_t<double>>().mutable_data();
                py::gil_scoped_release nogil;
                check(htool_hlu_debug_execute(s.p, first, last, factor.mutable_data(), diag.mutable_data(), rank.mutable_data(), norm0.mutable_data(), norm2.mutable_data(),
                                              counters.mutable_data(), r, ld_rhs, nrhs, sc));
            }, "first"_a, "last"_a, "factor"_a, "diag"_a, "rank"_a, "norm0"_a, "norm2"_a, "counters"_a, "rhs"_a = py::none(), "ld_rhs"_a = 0, "nrhs"_a = 0, "scratch"_a = py::none())
        .def("tables", [](const PyHluPlan &s) {
                const void *leaves, *diags;
                int64_t info[7];
                check(htool_hlu_plan_info(s.p, info, 7));
                check(htool_hlu_plan_tables(s.p, &leaves, &diags));
                py::array_t<uint8_t> l({(py::ssize_t)info[6], (py::ssize_t)48}); // (one record per rank slot)
                std::memcpy(l.mutable_data(), leaves, (size_t)info[6] * 48);
                py::array_t<uint8_t> d({(py::ssize_t)info[2], (py::ssize_t)24});
                std::memcpy(d.mutable_data(), diags, (size_t)info[2] * 24);
                return py::make_tuple(l, d);
            });

    declare_coefficient_classes<double>(m, "", "IGenerator", "VirtualGenerator", "VirtualLowRankGenerator", "NativeGenerator");
    declare_coefficient_classes<std::complex<double>>(m, "Complex", "IComplexGenerator", "ComplexVirtualGenerator", "VirtualComplexLowRankGenerator", "ComplexNativeGenerator");
}
