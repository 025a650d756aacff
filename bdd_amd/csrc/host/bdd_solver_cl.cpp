// bdd_solver_cl — command-line front end, as the reference's src/bdd_solver/bdd_solver_cl.cpp:
//     bdd_solver_cl <config.json | '{"input": "problem.lp", "relaxation solver": "cuda parallel mma", ...}'>
// Prints the final lower bound and, when "perturbation rounding" is configured, the primal objective.
#include <cstdio>
#include <exception>
#include <iostream>

#include "bdd_solver.hpp"

int main(int argc, char** argv)
{
    if (argc != 2) {
        std::cerr << "usage: " << argv[0] << " <config file | config json>\n";
        return 2;
    }
    try {
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
