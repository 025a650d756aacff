// bdd_solver_py.cpp — pybind11 module `bdd_solver_py`: the Python face of the C++ driver and of the GPU solver.
//
// Mirrors the reference's two pybind modules for this path:
//   bdd_solver_py.bdd_solver            src/bdd_solver/bdd_solver_py.cpp:9-20   (ctor from a config string / dict, solve,
//                                       lower_bound, min_marginals, min_marginals_with_variable_names)
//   bdd_cuda_parallel_mma_py.bdd_cuda_parallel_mma   src/bdd_solver/bdd_cuda_parallel_mma_py.cu:15-80  (ctor from an ILP, pickle
//                                       through the solver's own serialisation, sizes, lower_bound,
//                                       compute_and_set_min_marginal_diff on a device pointer)
// Everything computes through the C-ABI of include/bdd_mma.h / include/bdd_ilp.h; nothing here touches HIP.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <memory>
#include <stdexcept>
#include <string>
#include <unistd.h>
#include <vector>

#include "../../../include/bdd_ilp.h"
#include "../../../include/bdd_mma.h"
#include "bdd_solver.hpp"

namespace py = pybind11;

namespace {

std::string config_text(const py::object& cfg)
{
    if (py::isinstance<py::str>(cfg)) return cfg.cast<std::string>();
    return py::module_::import("json").attr("dumps")(cfg).cast<std::string>();  // a dict, as pybind11_json would take it
}

void ck(int rc, const bddmma_solver* s)
{
    if (rc != BDDMMA_OK) throw std::runtime_error(std::string("bdd_mma error ") + std::to_string(rc) + ": " + bddmma_last_error(s));
}

// LPMP::bdd_cuda_parallel_mma<REAL> as the Python module exposes it
struct hip_solver {
    bddmma_solver* h = nullptr;
    // what the solver's own bound leaves out: the ILP's constant and, for variables that occur in the objective only (any index), the
    // better of their two values — the same terms the bdd_solver driver adds, so both report the same bound for one ILP
    double constant = 0.0;
    hip_solver() = default;
    hip_solver(const hip_solver&) = delete;
    hip_solver& operator=(const hip_solver&) = delete;
    ~hip_solver() { bddmma_destroy(h); }

    // ctor from an ILP (bdd_cuda_parallel_mma_py.cu:38-43): .lp / OPB text -> one QBDD per row -> solver, objective as costs
    static std::unique_ptr<hip_solver> from_ilp(const std::string& text, const std::string& precision, int device)
    {
        bddilp* ilp = nullptr;
        if (bddilp_parse(text.c_str(), &ilp) != BDDILP_OK) throw std::runtime_error(std::string("cannot parse the ILP: ") + bddilp_last_error());
        bddilp_bdds* col = nullptr;
        if (bddilp_to_bdds(ilp, 0, 0, &col) != BDDILP_OK) {
            const std::string msg = bddilp_last_error();
            bddilp_destroy(ilp);
            throw std::runtime_error(msg);
        }
        std::vector<double> obj(bddilp_nr_variables(ilp));
        double constant = 0;
        bddilp_objective(ilp, obj.data(), &constant);
        const std::vector<double> full = obj;
        obj.resize(std::min<size_t>(obj.size(), bddilp_bdds_nr_variables(col)));
        auto s = std::make_unique<hip_solver>();
        const int prec = (precision == "float" || precision == "single") ? BDDMMA_F32 : BDDMMA_F64;
        const int rc = bddmma_create(&s->h, prec, device, bddilp_bdds_instructions(col), bddilp_bdds_delimiters(col), bddilp_bdds_nr_bdds(col),
                                     obj.data(), obj.size(), nullptr);
        bddilp_bdds_destroy(col);
        bddilp_destroy(ilp);
        ck(rc, nullptr);
        std::vector<int32_t> nb(bddmma_nr_variables(s->h), 0);
        ck(bddmma_num_bdds_per_var(s->h, nb.data()), s->h);
        s->constant = constant;
        for (size_t v = 0; v < full.size(); ++v)
            if ((v >= nb.size() || nb[v] == 0) && full[v] < 0) s->constant += full[v];
        return s;
    }
    // a scratch file for the archive: memory-backed when /dev/shm exists, else under TMPDIR or /tmp
    static std::string scratch_file()
    {
        const char* tmpdir = std::getenv("TMPDIR");
        for (const std::string& dir : {std::string("/dev/shm"), std::string(tmpdir ? tmpdir : "/tmp"), std::string("/tmp")}) {
            std::string path = dir + "/bddmma_pickle_XXXXXX";
            const int fd = mkstemp(&path[0]);
            if (fd >= 0) {
                close(fd);
                return path;
            }
        }
        throw std::runtime_error("cannot create a temporary file for (un)pickling");
    }
    // pickle (bdd_cuda_parallel_mma_py.cu:15-37): the solver's own archive (bddmma_save / bddmma_load) as bytes
    py::bytes dumps() const
    {
        const std::string path_s = scratch_file();
        const char* path = path_s.c_str();
        const int rc = bddmma_save(h, path);
        std::string blob;
        bool read_ok = false;
        if (rc == BDDMMA_OK) {
            std::ifstream f(path, std::ios::binary);
            if (f) {
                blob.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
                read_ok = !f.bad() && !blob.empty();
            }
        }
        std::remove(path);
        ck(rc, h);
        if (!read_ok) throw std::runtime_error("pickling: cannot read the solver archive back from " + path_s);
        return py::bytes(blob);
    }
    // (archive, constant, device): restored on the device it was on when that device exists in the unpickling process, else on device 0
    py::tuple getstate() const { return py::make_tuple(dumps(), constant, bddmma_device(h)); }
    static std::unique_ptr<hip_solver> setstate(const py::tuple& t)
    {
        if (t.size() != 3) throw std::runtime_error("bdd_hip_parallel_mma: invalid pickle state");
        int device = t[2].cast<int>();
        if (device < 0 || device >= bddmma_device_count()) device = 0;
        auto s = loads(t[0].cast<py::bytes>(), device);
        s->constant = t[1].cast<double>();
        return s;
    }
    static std::unique_ptr<hip_solver> loads(const py::bytes& b, int device)
    {
        const std::string blob = b;
        const std::string path_s = scratch_file();
        const char* path = path_s.c_str();
        {
            std::ofstream f(path, std::ios::binary);
            f.write(blob.data(), (std::streamsize)blob.size());
            f.flush();
            if (!f) { std::remove(path); throw std::runtime_error("unpickling: cannot write the solver archive to " + path_s); }
        }
        auto s = std::make_unique<hip_solver>();
        const int rc = bddmma_load(&s->h, device, path);
        std::remove(path);
        ck(rc, nullptr);
        return s;
    }
};

}  // namespace

PYBIND11_MODULE(bdd_solver_py, m)
{
    m.doc() = "Bindings for the BDD solver with the MI355X parallel-MMA backend.";
    using bddmma_host::bdd_solver;
    py::class_<bdd_solver>(m, "bdd_solver")
        .def(py::init([](const py::object& cfg, bool quiet) { return std::make_unique<bdd_solver>(config_text(cfg), quiet); }), py::arg("config"),
             py::arg("quiet") = false)
        .def("solve", [](py::object self) { self.cast<bdd_solver&>().solve(); return self; })
        .def("lower_bound", &bdd_solver::lower_bound)
        .def("min_marginals", &bdd_solver::min_marginals)   // [var][bdd] -> [mm0, mm1]
        .def("min_marginals_with_variable_names",           // bdd_solver.cpp:516-527: (names, mm0 per variable, mm1 per variable)
             [](bdd_solver& s) {
                 const auto mm = s.min_marginals();
                 std::vector<std::vector<double>> m0(mm.size()), m1(mm.size());
                 for (size_t v = 0; v < mm.size(); ++v)
                     for (const auto& p : mm[v]) { m0[v].push_back(p[0]); m1[v].push_back(p[1]); }
                 std::vector<std::string> names(s.ilp().var_names.begin(), s.ilp().var_names.begin() + (long)mm.size());
                 return py::make_tuple(names, m0, m1);
             })
        .def_property_readonly("solution",
                               [](const bdd_solver& s) -> py::object {
                                   if (s.solution().empty()) return py::none();
                                   py::list l;
                                   for (char c : s.solution()) l.append((int)c);
                                   return l;
                               })
        .def_property_readonly("solution_objective", &bdd_solver::solution_objective)
        .def_property_readonly("result", [](const bdd_solver& s) {
            const bddmma_run_result& r = s.result();
            py::dict d;
            d["iterations"] = r.iterations; d["lb_initial"] = r.lb_initial; d["lb_final"] = r.lb_final; d["seconds"] = r.seconds;
            d["stop_reason"] = r.stop_reason;
            return d;
        });

    py::class_<hip_solver>(m, "bdd_hip_parallel_mma")
        .def(py::init([](const std::string& ilp_text, const std::string& precision, int device) { return hip_solver::from_ilp(ilp_text, precision, device); }),
             py::arg("ilp"), py::arg("precision") = "double", py::arg("device") = 0)
        .def(py::pickle([](const hip_solver& s) { return s.getstate(); }, [](const py::tuple& t) { return hip_solver::setstate(t); }))
        .def_readonly("constant", &hip_solver::constant)  // ILP constant + better values of objective-only variables (part of lower_bound())
        .def("device", [](const hip_solver& s) { return bddmma_device(s.h); })
        .def("__repr__", [](const hip_solver& s) {
            return "<bdd_hip_parallel_mma>: nr_variables: " + std::to_string(bddmma_nr_variables(s.h)) + ", nr_bdds: " + std::to_string(bddmma_nr_bdds(s.h)) +
                   ", nr_layers: " + std::to_string(bddmma_nr_layers(s.h));
        })
        .def("nr_primal_variables", [](const hip_solver& s) { return bddmma_nr_variables(s.h); })
        .def("nr_layers", [](const hip_solver& s) { return bddmma_nr_layers(s.h); })
        // bdd_cuda_parallel_mma_py.cu:53: layers of one hop (bdd_cuda_base.h:106-109)
        .def("nr_layers", [](const hip_solver& s, int64_t hop_index) {
            std::vector<uint64_t> per_hop(bddmma_nr_hops(s.h));
            if (hop_index < 0 || (uint64_t)hop_index >= per_hop.size()) throw py::index_error("hop index out of range");
            ck(bddmma_layers_per_hop(s.h, per_hop.data()), s.h);
            return per_hop[(size_t)hop_index];
        }, py::arg("hop_index"))
        .def("nr_hops", [](const hip_solver& s) { return bddmma_nr_hops(s.h); })
        .def("nr_bdds", [](const hip_solver& s) { return bddmma_nr_bdds(s.h); })
        .def("iteration", [](hip_solver& s, double omega) { ck(bddmma_iteration(s.h, omega), s.h); }, py::arg("omega") = 0.5)
        .def("iterations", [](hip_solver& s, uint64_t n, double omega) { ck(bddmma_iterations(s.h, omega, n), s.h); }, py::arg("n"), py::arg("omega") = 0.5)
        .def("lower_bound", [](hip_solver& s) { double lb = 0; ck(bddmma_lower_bound(s.h, &lb), s.h); return lb + s.constant; })
        // bdd_cuda_parallel_mma_py.cu:56-72: hi - lo min-marginal of every layer into a caller-owned DEVICE buffer (REAL[nr_layers])
        .def("compute_and_set_min_marginal_diff",
             [](hip_solver& s, uint64_t mm_diff_out_ptr) { ck(bddmma_min_marginal_diff(s.h, reinterpret_cast<void*>(mm_diff_out_ptr), 1), s.h); })
        // the same into a new host list (no counterpart in the reference; convenient without a device buffer)
        .def("min_marginal_diff", [](hip_solver& s) {
            const size_t L = bddmma_nr_layers(s.h);
            std::vector<double> out(L);
            if (bddmma_precision(s.h) == BDDMMA_F64) {
                ck(bddmma_min_marginal_diff(s.h, out.data(), 0), s.h);
            } else {
                std::vector<float> f(L);
                ck(bddmma_min_marginal_diff(s.h, f.data(), 0), s.h);
                out.assign(f.begin(), f.end());
            }
            return out;
        });
}
