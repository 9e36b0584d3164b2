// A minimal reader for the netCDF-4 (HDF5) files of the reference's `in_fmt SeaNetCDF` inputs (data/3D/large_tests/white_sea):
// root group with compact links -> datasets with a version-2 object header, IEEE little-endian floats, contiguous or chunked
// layout (version-1 B-tree index), deflate / shuffle filters.  That is what the netCDF-4 library writes for such a file;
// anything else throws.  The reference reads these files through libnetcdf (Grid3D::LoadNetCDF, FluidSolver3D/Grid3D.cpp:433-486);
// neither libnetcdf nor libhdf5 exists in this image -- zlib does.  Format: HDF5 File Format Specification version 3.0.
// Python twin: cmc_fluid_solver_amd/hdf5_min.py.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace fs3d {

class Hdf5File {
    std::vector<unsigned char> d_;
    uint64_t base_ = 0;
    std::map<std::string, uint64_t> links_;
    struct Msg { int type; size_t pos, size; };
    static constexpr uint64_t UNDEF = ~0ull;

    uint64_t u(size_t p, int n) const
    {
        if (p + n > d_.size()) throw std::runtime_error("HDF5: read past the end of the file");
        uint64_t v = 0;
        for (int i = n - 1; i >= 0; i--) v = (v << 8) | d_[p + i];
        return v;
    }
    bool sig(size_t p, const char *s) const { return p + 4 <= d_.size() && std::memcmp(&d_[p], s, 4) == 0; }

    void walk(size_t p, size_t end, int flags, std::vector<Msg> &out, int depth = 0) const
    {
        if (end > d_.size() || depth > 64) throw std::runtime_error("HDF5: object header block past the end of the file");
        while (p + 4 <= end) {
            const int t = d_[p];
            const size_t sz = (size_t)u(p + 1, 2);
            p += 4 + ((flags & 0x04) ? 2 : 0);
            if (t == 0x10) {                                      // continuation
                const size_t ca = (size_t)(u(p, 8) + base_), cl = (size_t)u(p + 8, 8);
                if (!sig(ca, "OCHK")) throw std::runtime_error("HDF5: bad object header continuation");
                if (cl < 8) throw std::runtime_error("HDF5: bad object header continuation");
                walk(ca + 4, ca + cl - 4, flags, out, depth + 1);
            } else if (t != 0) { if (p + sz > end) throw std::runtime_error("HDF5: message past its header block"); out.push_back({t, p, sz}); }
            p += sz;
        }
    }
    std::vector<Msg> messages(uint64_t addr) const
    {
        size_t a = (size_t)(addr + base_);
        if (!sig(a, "OHDR") || u(a + 4, 1) != 2) throw std::runtime_error("HDF5: object header version 2 expected");
        const int flags = (int)u(a + 5, 1);
        size_t p = a + 6;
        if (flags & 0x20) p += 16;
        if (flags & 0x10) p += 4;
        const int szf = 1 << (flags & 3);
        const size_t chunk0 = (size_t)u(p, szf);
        p += szf;
        std::vector<Msg> out;
        walk(p, p + chunk0, flags, out);
        return out;
    }

    struct Info { std::vector<uint64_t> dims; int esize = 0; Msg layout{0, 0, 0}; std::vector<int> filters; };
    Info info(const std::string &name) const
    {
        auto it = links_.find(name);
        if (it == links_.end()) throw std::runtime_error("HDF5: no dataset " + name + " in the root group");
        Info r;
        for (const Msg &m : messages(it->second)) {
            const size_t b = m.pos;
            if (m.type == 0x01) {                                 // dataspace
                const int ver = d_[b], rank = d_[b + 1];
                const size_t p = b + (ver == 2 ? 4 : 8);
                if (ver != 1 && ver != 2) throw std::runtime_error("HDF5: dataspace message version");
                for (int k = 0; k < rank; k++) r.dims.push_back(u(p + 8 * k, 8));
            } else if (m.type == 0x03) {                          // datatype
                const int cls = d_[b] & 0x0F;
                r.esize = (int)u(b + 4, 4);
                if (cls != 1 || (d_[b + 1] & 1) || (r.esize != 4 && r.esize != 8)) throw std::runtime_error("HDF5: only little-endian IEEE float32 / float64 datasets are supported");
            } else if (m.type == 0x0B) {                          // filter pipeline
                if (d_[b] != 2) throw std::runtime_error("HDF5: filter pipeline message version");
                size_t p = b + 2;
                for (int k = 0; k < d_[b + 1]; k++) {
                    const int fid = (int)u(p, 2); p += 2;
                    size_t nl = 0;
                    if (fid >= 256) { nl = (size_t)u(p, 2); p += 2; }
                    p += 2;
                    const size_t ncv = (size_t)u(p, 2); p += 2 + nl + 4 * ncv;
                    if (fid != 1 && fid != 2) throw std::runtime_error("HDF5: only the shuffle and deflate filters are supported");
                    r.filters.push_back(fid);
                }
            } else if (m.type == 0x08) {
                if (d_[b] != 3) throw std::runtime_error("HDF5: data layout message version");
                r.layout = m;
            }
        }
        if (r.dims.empty() || !r.esize || !r.layout.size) throw std::runtime_error("HDF5: dataset " + name + ": incomplete object header");
        return r;
    }

    struct Chunk { std::vector<uint64_t> offs; uint64_t addr; uint32_t size, mask; };
    // B-tree of the raw data chunks.  Level and child addresses come from the file: the walk is bounded (a child must be one
    // level below its parent, at most 8 levels, entries must fit the file) so that a node that names itself cannot recurse for ever.
    void chunks(uint64_t addr, int rank, std::vector<Chunk> &out, int parent_level = -1, int depth = 0) const
    {
        size_t p = (size_t)(addr + base_);
        if (depth > 8 || !sig(p, "TREE") || u(p + 4, 1) != 1) throw std::runtime_error("HDF5: version-1 B-tree of raw data chunks expected");
        const int level = (int)u(p + 5, 1), n = (int)u(p + 6, 2);
        if (parent_level >= 0 && level != parent_level - 1) throw std::runtime_error("HDF5: B-tree child is not one level below its parent");
        p += 24;
        const size_t ks = 8 + 8 * (size_t)(rank + 1);
        if (p + (size_t)n * (ks + 8) > d_.size()) throw std::runtime_error("HDF5: B-tree node past the end of the file");
        for (int e = 0; e < n; e++) {
            Chunk c;
            c.size = (uint32_t)u(p, 4); c.mask = (uint32_t)u(p + 4, 4);
            for (int k = 0; k < rank; k++) c.offs.push_back(u(p + 8 + 8 * k, 8));
            c.addr = u(p + ks, 8);
            p += ks + 8;
            if (level == 0) out.push_back(c); else chunks(c.addr, rank, out, level, depth + 1);
        }
    }

public:
    explicit Hdf5File(const std::string &path)
    {
        std::ifstream in(path.c_str(), std::ios::binary);
        if (!in) throw std::runtime_error("cannot open " + path);
        d_.assign(std::istreambuf_iterator<char>(in), std::istreambuf_iterator<char>());
        static const unsigned char magic[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
        if (d_.size() < 48 || std::memcmp(d_.data(), magic, 8) != 0) throw std::runtime_error(path + ": not an HDF5 (netCDF-4) file");
        if ((d_[8] != 2 && d_[8] != 3) || d_[9] != 8 || d_[10] != 8) throw std::runtime_error("HDF5: superblock version / offset size not supported");
        base_ = u(12, 8);
        const uint64_t root = u(36, 8);
        for (const Msg &m : messages(root)) {
            if (m.type != 0x06) continue;                         // link
            size_t p = m.pos;
            if (d_[p] != 1) throw std::runtime_error("HDF5: link message version");
            const int fl = d_[p + 1];
            p += 2;
            int ltype = 0;
            if (fl & 0x08) ltype = d_[p++];
            if (fl & 0x04) p += 8;
            if (fl & 0x10) p += 1;
            const int n = 1 << (fl & 3);
            const size_t ln = (size_t)u(p, n); p += n;
            if (p + ln + 8 > d_.size()) throw std::runtime_error("HDF5: link name past the end of the file");
            const std::string name((const char *)&d_[p], ln); p += ln;
            if (ltype == 0) links_[name] = u(p, 8);
        }
        if (links_.empty()) throw std::runtime_error("HDF5: no compact links in the root group (dense link storage is not supported)");
    }

    std::vector<uint64_t> Shape(const std::string &name) const { return info(name).dims; }

    // the whole dataset, row-major, converted to double
    std::vector<double> Read(const std::string &name) const
    {
        const Info r = info(name);
        const int rank = (int)r.dims.size();
        size_t total = 1;
        for (uint64_t n : r.dims) {
            if (n == 0 || n > ((uint64_t)1 << 28) || total > ((size_t)1 << 28) / (size_t)n) throw std::runtime_error("HDF5: dataset too large for this reader (or a corrupt dataspace)");
            total *= (size_t)n;
        }
        std::vector<double> out(total, 0.0);
        auto value = [&](const unsigned char *p) { if (r.esize == 4) { float f; std::memcpy(&f, p, 4); return (double)f; } double v; std::memcpy(&v, p, 8); return v; };
        const size_t L = r.layout.pos;
        const int cls = d_[L + 1];
        if (cls == 1) {                                           // contiguous
            const size_t a = (size_t)(u(L + 2, 8) + base_);
            if (a + total * r.esize > d_.size()) throw std::runtime_error("HDF5: contiguous data past the end of the file");
            for (size_t i = 0; i < total; i++) out[i] = value(&d_[a + i * r.esize]);
            return out;
        }
        if (cls != 2 || d_[L + 2] != rank + 1 || rank > 3) throw std::runtime_error("HDF5: data layout not supported");
        const uint64_t btree = u(L + 3, 8);
        std::vector<uint64_t> cd(rank);
        size_t celems = 1;
        for (int k = 0; k < rank; k++) {
            cd[k] = u(L + 11 + 4 * k, 4);
            if (cd[k] == 0 || celems > ((size_t)1 << 28) / (size_t)cd[k]) throw std::runtime_error("HDF5: chunk too large for this reader (or a corrupt layout)");
            celems *= (size_t)cd[k];
        }
        if (btree == UNDEF) return out;
        std::vector<Chunk> cs;
        chunks(btree, rank, cs);
        for (const Chunk &c : cs) {
            if (c.addr + base_ > d_.size() || c.size > d_.size() - (size_t)(c.addr + base_)) throw std::runtime_error("HDF5: chunk data past the end of the file");
            std::vector<unsigned char> raw(d_.begin() + (size_t)(c.addr + base_), d_.begin() + (size_t)(c.addr + base_) + c.size);
            for (int k = (int)r.filters.size() - 1; k >= 0; k--) {     // undo the pipeline back to front
                if (c.mask & (1u << k)) continue;
                if (r.filters[k] == 1) {
                    std::vector<unsigned char> o(celems * r.esize);
                    uLongf n = (uLongf)o.size();
                    if (uncompress(o.data(), &n, raw.data(), (uLong)raw.size()) != Z_OK) throw std::runtime_error("HDF5: inflate failed");
                    o.resize(n); raw.swap(o);
                } else {
                    std::vector<unsigned char> o(raw.size());
                    const size_t ne = raw.size() / r.esize;
                    for (int b = 0; b < r.esize; b++) for (size_t e = 0; e < ne; e++) o[e * r.esize + b] = raw[(size_t)b * ne + e];
                    raw.swap(o);
                }
            }
            if (raw.size() != celems * r.esize) throw std::runtime_error("HDF5: chunk size mismatch");
            // copy the part of the chunk that lies inside the dataset (rank <= 3)
            uint64_t n[3] = {1, 1, 1}, cdim[3] = {1, 1, 1}, o[3] = {0, 0, 0};
            for (int k = 0; k < rank; k++) { n[3 - rank + k] = r.dims[k]; cdim[3 - rank + k] = cd[k]; o[3 - rank + k] = c.offs[k]; }
            for (uint64_t a = 0; a < cdim[0] && o[0] + a < n[0]; a++)
                for (uint64_t b = 0; b < cdim[1] && o[1] + b < n[1]; b++)
                    for (uint64_t e = 0; e < cdim[2] && o[2] + e < n[2]; e++)
                        out[(size_t)(((o[0] + a) * n[1] + o[1] + b) * n[2] + o[2] + e)] = value(&raw[(size_t)((a * cdim[1] + b) * cdim[2] + e) * r.esize]);
        }
        return out;
    }
};

}  // namespace fs3d
