// bdd_solver_cl — command-line front end, as the reference's src/bdd_solver/bdd_solver_cl.cpp:
//     bdd_solver_cl <config.json | '{"input": "problem.lp", "relaxation solver": "cuda parallel mma", ...}'>
// prints the final lower bound and, when "perturbation rounding" is configured, the primal objective.
// Batch farm over the GPUs of a node (independent instances, one host thread per device, no collective):
//     bdd_solver_cl --batch cfg1.json cfg2.json ... [--devices 0-7 | 0,2,5] [--quiet]
//     bdd_solver_cl --bench-set-cover V B k --iterations N [--warmup W] [--seeds 12345-12352] [--devices 0-7] [--precision float]
// Both print one JSON line per instance; the benchmark adds the aggregate iterations/s (all instances / slowest instance's time).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <iostream>
#include <string>
#include <vector>

#include "../../../include/bdd_mma.h"
#include "bdd_solver.hpp"

namespace {
// "0-7", "0,2,5", "3"  ->  list of integers
std::vector<uint64_t> parse_list(const std::string& s)
{
    std::vector<uint64_t> out;
    size_t pos = 0;
    while (pos < s.size()) {
        size_t end = s.find(',', pos);
        if (end == std::string::npos) end = s.size();
        const std::string tok = s.substr(pos, end - pos);
        const size_t dash = tok.find('-');
        if (dash != std::string::npos && dash > 0) {
            const uint64_t a = std::stoull(tok.substr(0, dash)), b = std::stoull(tok.substr(dash + 1));
            for (uint64_t v = a; v <= b; ++v) out.push_back(v);
        } else if (!tok.empty()) {
            out.push_back(std::stoull(tok));
        }
        pos = end + 1;
    }
    return out;
}
std::vector<int> all_devices()
{
    std::vector<int> d;
    for (int i = 0; i < bddmma_device_count(); ++i) d.push_back(i);
    return d;
}
std::string json_escape(const std::string& s)
{
    std::string o;
    for (char c : s) {
        if (c == '"' || c == '\\') { o += '\\'; o += c; }
        else if (c == '\n') o += "\\n";
        else o += c;
    }
    return o;
}
}  // namespace

int main(int argc, char** argv)
{
    try {
        std::vector<std::string> args(argv + 1, argv + argc);
        if (!args.empty() && (args[0] == "--batch" || args[0] == "--bench-set-cover")) {
            std::vector<int> devices;
            std::vector<std::string> pos;
            std::vector<uint64_t> seeds;
            std::string precision = "float";
            uint64_t iterations = 1000, warmup = 100;
            bool quiet = false;
            for (size_t i = 1; i < args.size(); ++i) {
                auto need = [&](const char* what) -> const std::string& {
                    if (i + 1 >= args.size()) throw std::runtime_error(std::string(what) + " needs a value");
                    return args[++i];
                };
                if (args[i] == "--devices") { for (uint64_t d : parse_list(need("--devices"))) devices.push_back((int)d); }
                else if (args[i] == "--seeds") seeds = parse_list(need("--seeds"));
                else if (args[i] == "--precision") precision = need("--precision");
                else if (args[i] == "--iterations") iterations = std::stoull(need("--iterations"));
                else if (args[i] == "--warmup") warmup = std::stoull(need("--warmup"));
                else if (args[i] == "--quiet") quiet = true;
                else pos.push_back(args[i]);
            }
            if (devices.empty()) devices = all_devices();
            if (devices.empty()) throw std::runtime_error("no HIP device available (this backend has no CPU fallback)");
            if (args[0] == "--batch") {
                if (pos.empty()) throw std::runtime_error("--batch needs at least one config");
                int failed = 0;
                for (const auto& r : bddmma_host::solve_batch(pos, devices, quiet)) {
                    std::printf("{\"config\": \"%s\", \"device\": %d, \"ok\": %s, \"lower_bound\": %.12g, \"iterations\": %llu, \"seconds\": %.4f", json_escape(r.config).c_str(),
                                r.device, r.ok ? "true" : "false", r.lower_bound, (unsigned long long)r.iterations, r.seconds);
                    if (r.has_primal) std::printf(", \"primal\": %.12g", r.primal);
                    if (!r.ok) std::printf(", \"error\": \"%s\"", json_escape(r.error).c_str());
                    std::printf("}\n");
                    failed += r.ok ? 0 : 1;
                }
                return failed ? 1 : 0;
            }
            if (pos.size() != 3) throw std::runtime_error("--bench-set-cover needs V B k");
            if (seeds.empty()) for (size_t i = 0; i < devices.size(); ++i) seeds.push_back(12345 + i);
            double aggregate = 0;
            int failed = 0;
            for (const auto& r : bddmma_host::bench_set_cover(std::stoull(pos[0]), std::stoull(pos[1]), std::stoull(pos[2]), seeds, devices, precision, warmup,
                                                              iterations, &aggregate)) {
                std::printf("{\"device\": %d, \"seed\": %llu, \"ok\": %s, \"construct_seconds\": %.4f, \"iterations_per_second\": %.2f, \"lower_bound\": %.12g", r.device,
                            (unsigned long long)r.seed, r.ok ? "true" : "false", r.construct_seconds, r.iterations_per_second, r.lower_bound);
                if (!r.ok) std::printf(", \"error\": \"%s\"", json_escape(r.error).c_str());
                std::printf("}\n");
                failed += r.ok ? 0 : 1;
            }
            std::printf("{\"instances\": %zu, \"devices\": %zu, \"iterations\": %llu, \"precision\": \"%s\", \"aggregate_iterations_per_second\": %.2f}\n", seeds.size(),
                        devices.size(), (unsigned long long)iterations, precision.c_str(), aggregate);
            return failed ? 1 : 0;
        }
        if (argc != 2) {
            std::cerr << "usage: " << argv[0] << " <config file | config json>\n       " << argv[0] << " --batch cfg... [--devices 0-7] [--quiet]\n       " << argv[0]
                      << " --bench-set-cover V B k --iterations N [--warmup W] [--seeds a-b] [--devices 0-7] [--precision float|double]\n";
            return 2;
        }
        bddmma_host::bdd_solver solver(argv[1]);
        solver.solve();
        std::printf("[bdd solver] final lower bound = %.12g\n", solver.lower_bound());
        if (!solver.solution().empty()) std::printf("[bdd solver] primal objective = %.12g\n", solver.solution_objective());
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
