// The reference's run configuration (Common/Config.h:76-271): the same keys, defaults and checks.
//   * whitespace-separated `key value` tokens, unknown keys are skipped token by token (Config.h:195-245);
//   * every real number goes through "%f" into a float and is widened to double (ReadDouble, :116-135) --
//     `grid_dx 0.02` is 0.0199999995529651641845703125, which decides grid dimensions (ceil(len/dx) + 1);
//   * where the reference prints a message and calls exit(0) (:249-270) this throws std::runtime_error.
// CRLF files are accepted.  Python twin: cmc_fluid_solver_amd/shape2d.py (class Config), same tests.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace fs3d {

struct Config {
    // defaults: Config.h:76-113
    double R_specific = 461.495, k = 0.6, cv = 4200.0, baseT = 1.0;
    bool bc_noslip = true;
    double bc_strength = 0.5, bc_inV[3] = {0, 0, 0}, bc_inT = 1.0;
    bool useNormalizedParams = false;
    double viscosity = 0.05, density = 1000.0, Re = -1, Pr = -1, lambda = -1;
    double depth_var = 0.0;
    int cycles = 1, time_steps = 50, out_time_steps = 10, outdimx = 50, outdimy = 50, outdimz = 50;
    std::vector<std::string> out_vars;
    int num_global = 2, num_local = 1;
    std::string problem_dim, in_fmt, out_fmt, solver;
    double frame_time = -1, dx = -1, dy = -1, dz = -1, depth = -1;

    static double real(const std::string &tok)
    {
        float f = 0.0f;
        std::sscanf(tok.c_str(), "%f", &f);      // ReadDouble: fscanf("%f") into a float, then widened
        return (double)f;
    }

    void Load(const std::string &path)
    {
        std::ifstream in(path.c_str());
        if (!in) throw std::runtime_error("cannot open config file " + path);
        std::vector<std::string> t;
        for (std::string w; in >> w;) t.push_back(w);       // operator>> splits on any whitespace incl. '\r'
        auto need = [&](size_t i) { if (i >= t.size()) throw std::runtime_error("config: value missing after " + t[i - 1]); };
        for (size_t i = 0; i < t.size();) {
            const std::string key = t[i++];
            auto rd = [&](double &dst) { need(i); dst = real(t[i++]); };
            auto ri = [&](int &dst) { need(i); dst = std::atoi(t[i++].c_str()); };
            if (key == "viscosity") rd(viscosity);
            else if (key == "density") rd(density);
            else if (key == "Re") { rd(Re); useNormalizedParams = true; }
            else if (key == "Pr") { rd(Pr); useNormalizedParams = true; }
            else if (key == "lambda") { rd(lambda); useNormalizedParams = true; }
            else if (key == "bc_strenght") rd(bc_strength);                      // sic, Config.h:214
            else if (key == "bc_initT") rd(bc_inT);
            else if (key == "bc_initv") { for (int c = 0; c < 3; c++) rd(bc_inV[c]); }
            else if (key == "bc_type") { need(i); bc_noslip = t[i][0] == 'N' || t[i][0] == 'n'; i++; }
            else if (key == "grid_dx") rd(dx);
            else if (key == "grid_dy") rd(dy);
            else if (key == "grid_dz") rd(dz);
            else if (key == "frame_time") rd(frame_time);
            else if (key == "depth") rd(depth);
            else if (key == "depth_var") rd(depth_var);
            else if (key == "cycles") ri(cycles);
            else if (key == "time_steps") ri(time_steps);
            else if (key == "out_time_steps") ri(out_time_steps);
            else if (key == "out_gridx") ri(outdimx);
            else if (key == "out_gridy") ri(outdimy);
            else if (key == "out_gridz") ri(outdimz);
            else if (key == "num_global") ri(num_global);
            else if (key == "num_local") ri(num_local);
            else if (key == "dimension") { need(i); problem_dim = t[i++]; }
            else if (key == "in_fmt") { need(i); in_fmt = t[i++]; }
            else if (key == "out_fmt") { need(i); out_fmt = t[i++]; }
            else if (key == "solver") { need(i); solver = t[i++]; }
            else if (key == "out_vars") { int n = 0; ri(n); out_vars.clear(); for (int c = 0; c < n; c++) { need(i); out_vars.push_back(t[i++]); } }
        }
        // Config.h:249-270
        if (problem_dim.empty()) throw std::runtime_error("must specify problem dimension!");
        if (solver.empty()) throw std::runtime_error("must specify solver!");
        if (out_fmt.empty()) throw std::runtime_error("must specify output format!");
        if (dx < 0) throw std::runtime_error("cannot find dx!");
        if (dy < 0) throw std::runtime_error("cannot find dy!");
        if (problem_dim == "3D") {
            if (out_vars.empty()) throw std::runtime_error("must output at least 1 var!");
            if (in_fmt.empty()) throw std::runtime_error("must specify input format!");
            if (dz < 0) throw std::runtime_error("cannot find dz!");
            if (in_fmt == "Shape2D" && depth < 0) throw std::runtime_error("cannot find depth!");
        }
        if (useNormalizedParams && (Re < 0 || Pr < 0 || lambda < 0)) throw std::runtime_error("must specify Re, Pr and lambda!");
    }
};

}  // namespace fs3d
