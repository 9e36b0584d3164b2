// Result file of a 3D run: the reference's OutputNetCDF3D_header / OutputNetCDF3D_layer (Common/IO.h:136-276, 350-388)
// -- dimensions x, y, z, t (unlimited); coordinate variables x, y, z (float) and time (double); u, v, w, T as
// double (t, x, y, z) as chosen by `out_vars`; the same attributes (units, actual_range, valid_range,
// missing_value 99999, long_name, var_desc; global Conventions/title/history/description/platform).
// The reference links libnetcdf and creates an NC_NETCDF4 (HDF5) file; neither library exists here, so this writes
// the netCDF *classic* format, 64-bit-offset variant (CDF-2), by hand -- the same data model, read by ncdump,
// ncview, netCDF4-python and scipy.io.netcdf_file alike.  Difference kept on purpose: `time` is a record variable
// here and receives its value when a layer is appended (the reference pre-fills all time values in the header call).
// The optional depth variable `d` (SeaNetCDF inputs: float (x, y), the depth map resampled to the output grid) is a fixed-size
// variable written with the header, as in the reference.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace fs3d {

class NetCDF3Writer {
    enum { NC_CHAR = 2, NC_FLOAT = 5, NC_DOUBLE = 6, NC_DIMENSION = 10, NC_VARIABLE = 11, NC_ATTRIBUTE = 12 };
    struct Att { std::string name; int type; std::vector<unsigned char> data; int nelems; };
    struct Var { std::string name; int type; std::vector<int> dims; std::vector<Att> atts; bool record; uint64_t vsize = 0, begin = 0; };
    std::vector<unsigned char> h_;
    std::string path_;
    std::vector<Var> vars_;
    uint64_t recsize_ = 0, rec_begin_ = 0;
    uint32_t numrecs_ = 0;
    int outx_ = 0, outy_ = 0, outz_ = 0;
    double timestep_ = 0;

    static void be(std::vector<unsigned char> &b, const void *p, int n) { const unsigned char *c = (const unsigned char *)p; for (int i = n - 1; i >= 0; i--) b.push_back(c[i]); }
    static void u32(std::vector<unsigned char> &b, uint32_t v) { be(b, &v, 4); }
    static void u64(std::vector<unsigned char> &b, uint64_t v) { be(b, &v, 8); }
    static void pad4(std::vector<unsigned char> &b) { while (b.size() % 4) b.push_back(0); }
    static void name(std::vector<unsigned char> &b, const std::string &s) { u32(b, (uint32_t)s.size()); b.insert(b.end(), s.begin(), s.end()); pad4(b); }
    static Att text(const std::string &n, const std::string &v) { Att a{n, NC_CHAR, {}, (int)v.size()}; a.data.assign(v.begin(), v.end()); return a; }
    static Att floats(const std::string &n, const float *v, int k) { Att a{n, NC_FLOAT, {}, k}; for (int i = 0; i < k; i++) be(a.data, &v[i], 4); return a; }
    static Att doubles(const std::string &n, const double *v, int k) { Att a{n, NC_DOUBLE, {}, k}; for (int i = 0; i < k; i++) be(a.data, &v[i], 8); return a; }
    static void atts(std::vector<unsigned char> &b, const std::vector<Att> &l)
    {
        if (l.empty()) { u32(b, 0); u32(b, 0); return; }
        u32(b, NC_ATTRIBUTE); u32(b, (uint32_t)l.size());
        for (const Att &a : l) { name(b, a.name); u32(b, (uint32_t)a.type); u32(b, (uint32_t)a.nelems); b.insert(b.end(), a.data.begin(), a.data.end()); pad4(b); }
    }
    std::vector<unsigned char> header(const std::vector<uint32_t> &dimlen, const std::vector<std::string> &dimname, const std::vector<Att> &gatts) const
    {
        std::vector<unsigned char> b{'C', 'D', 'F', 2};
        u32(b, numrecs_);
        u32(b, NC_DIMENSION); u32(b, (uint32_t)dimlen.size());
        for (size_t i = 0; i < dimlen.size(); i++) { name(b, dimname[i]); u32(b, dimlen[i]); }
        atts(b, gatts);
        u32(b, NC_VARIABLE); u32(b, (uint32_t)vars_.size());
        for (const Var &v : vars_) {
            name(b, v.name); u32(b, (uint32_t)v.dims.size());
            for (int d : v.dims) u32(b, (uint32_t)d);
            atts(b, v.atts); u32(b, (uint32_t)v.type); u32(b, (uint32_t)v.vsize); u64(b, v.begin);
        }
        return b;
    }

public:
    // OutputNetCDF3D_header: bbox = {xmin, ymin, zmin, xmax, ymax, zmax}; timestep = dt * out_time_steps; time = final time
    void Create(const std::string &path, const float bbox[6], double timestep, double time, int outdimx, int outdimy, int outdimz,
                const std::vector<std::string> &vars, bool xy_degree_units = false, const float *depth_xy = nullptr)
    {
        path_ = path; outx_ = outdimx; outy_ = outdimy; outz_ = outdimz; timestep_ = timestep; numrecs_ = 0;
        const std::vector<std::string> dimname{"x", "y", "z", "t"};
        const std::vector<uint32_t> dimlen{(uint32_t)outdimx, (uint32_t)outdimy, (uint32_t)outdimz, 0u};   // 0 = the record dimension
        vars_.clear();
        const char *axis[3] = {"x", "y", "z"};
        for (int a = 0; a < 3; a++) {
            Var v{axis[a], NC_FLOAT, {a}, {}, false};
            const float rng[2] = {bbox[a], bbox[3 + a]};
            v.atts.push_back(floats("actual_range", rng, 2));
            v.atts.push_back(text("long_name", std::string(axis[a]) + " coord"));
            v.atts.push_back(text("units", a == 2 ? "metres" : (xy_degree_units ? (a == 0 ? "degree_north" : "degree_east") : "metres")));
            vars_.push_back(v);
        }
        {
            Var v{"time", NC_DOUBLE, {3}, {}, true};
            const double tt[2] = {0.0, time};
            v.atts.push_back(text("units", "s")); v.atts.push_back(doubles("actual_range", tt, 2)); v.atts.push_back(text("long_name", "time"));
            vars_.push_back(v);
        }
        const char *vshort[4] = {"u", "v", "w", "T"}, *vlong[4] = {"x-velocity", "y-velocity", "z-velocity", "temperature"};
        for (int i = 0; i < 4; i++) {
            bool use = false;
            for (const auto &s : vars) use = use || s == vshort[i];
            if (!use) continue;
            Var v{vshort[i], NC_DOUBLE, {3, 0, 1, 2}, {}, true};
            const double rng[2] = {-1.0, 1.0};
            const float miss = 99999.0f;                           // MISSING_VALUE, Geometry.h
            v.atts.push_back(text("units", i == 3 ? "tmp" : "m/s"));
            v.atts.push_back(doubles("actual_range", rng, 2)); v.atts.push_back(doubles("valid_range", rng, 2));
            v.atts.push_back(floats("missing_value", &miss, 1));
            v.atts.push_back(text("long_name", vlong[i])); v.atts.push_back(text("var_desc", vshort[i]));
            vars_.push_back(v);
        }
        bool use_d = false;
        for (const auto &s : vars) use_d = use_d || s == "d";
        if (use_d) {
            // the reference dereferences a null DepthInfo3D for inputs that carry no depth map (IO.h:270-273); here it is an error
            if (!depth_xy) throw std::runtime_error("out_vars: the depth variable `d` needs a SeaNetCDF input");
            Var v{"d", NC_FLOAT, {0, 1}, {}, false};
            const double rng[2] = {-1.0, 1.0};
            const float miss = 99999.0f;
            v.atts.push_back(text("units", "m"));
            v.atts.push_back(doubles("actual_range", rng, 2)); v.atts.push_back(doubles("valid_range", rng, 2));
            v.atts.push_back(floats("missing_value", &miss, 1));
            v.atts.push_back(text("long_name", "depth")); v.atts.push_back(text("var_desc", "d"));
            vars_.push_back(v);
        }
        const std::vector<Att> gatts{text("Conventions", "COARDS"), text("title", "cmc-fluid-solver results"),
                                     text("history", "created by using cmc-fluid-solver"), text("description", "Test data"), text("platform", "Model")};
        // sizes and offsets: fixed variables first, then the records
        const uint64_t cells = (uint64_t)outdimx * outdimy * outdimz;
        if (cells * 8 >= 0xFFFFFFFCull) throw std::runtime_error("output grid too large for one netCDF classic record variable (>= 4 GiB per layer)");
        for (Var &v : vars_) v.vsize = v.record ? (v.dims.size() == 1 ? 8 : cells * 8) : (v.dims.size() == 2 ? (uint64_t)outdimx * outdimy * 4 : (uint64_t)dimlen[v.dims[0]] * 4);
        for (Var &v : vars_) v.vsize = (v.vsize + 3) / 4 * 4;
        uint64_t off = header(dimlen, dimname, gatts).size();
        for (Var &v : vars_) if (!v.record) { v.begin = off; off += v.vsize; }
        rec_begin_ = off; recsize_ = 0;
        for (Var &v : vars_) if (v.record) { v.begin = off; off += v.vsize; recsize_ += v.vsize; }
        h_ = header(dimlen, dimname, gatts);
        FILE *f = std::fopen(path.c_str(), "wb");
        if (!f) throw std::runtime_error("cannot create " + path);
        std::fwrite(h_.data(), 1, h_.size(), f);
        // axis data: pMin + dd * i with dd = (pMax - pMin) / outdim   (IO.h:241-258)
        const int od[3] = {outdimx, outdimy, outdimz};
        for (int a = 0; a < 3; a++) {
            const float dd = (float)(bbox[3 + a] - bbox[a]) / (od[a]);
            std::vector<unsigned char> b;
            for (int i = 0; i < od[a]; i++) { const float x = bbox[a] + dd * i; be(b, &x, 4); }
            std::fwrite(b.data(), 1, b.size(), f);
        }
        if (use_d) {                                           // after x, y, z: the order of vars_ (fixed-size variables in definition order)
            std::vector<unsigned char> b;
            for (size_t c = 0; c < (size_t)outdimx * outdimy; c++) be(b, &depth_xy[c], 4);
            std::fwrite(b.data(), 1, b.size(), f);
        }
        std::fclose(f);
    }
    // OutputNetCDF3D_layer: vel = interleaved x,y,z per cell (Vec3D), T = double, both [outdimx][outdimy][outdimz]
    template <typename FTYPE>
    void AppendLayer(const FTYPE *vel, const double *T)
    {
        FILE *f = std::fopen(path_.c_str(), "r+b");
        if (!f) throw std::runtime_error("cannot reopen " + path_);
        const uint64_t cells = (uint64_t)outx_ * outy_ * outz_;
        std::vector<unsigned char> rec;
        rec.reserve((size_t)recsize_);
        for (const Var &v : vars_) {
            if (!v.record) continue;
            if (v.dims.size() == 1) { const double t = numrecs_ * timestep_; be(rec, &t, 8); continue; }
            const int comp = v.name == "u" ? 0 : (v.name == "v" ? 1 : (v.name == "w" ? 2 : 3));
            for (uint64_t c = 0; c < cells; c++) { const double x = comp == 3 ? T[c] : (double)vel[3 * c + comp]; be(rec, &x, 8); }
        }
        std::fseek(f, (long)(rec_begin_ + (uint64_t)numrecs_ * recsize_), SEEK_SET);
        std::fwrite(rec.data(), 1, rec.size(), f);
        numrecs_++;
        std::vector<unsigned char> n;
        u32(n, numrecs_);
        std::fseek(f, 4, SEEK_SET);
        std::fwrite(n.data(), 1, 4, f);
        std::fclose(f);
    }
    uint32_t NumRecords() const { return numrecs_; }
};

}  // namespace fs3d
