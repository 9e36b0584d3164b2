// Grid3D::OutputImage (FluidSolver3D/Grid3D.cpp:1112-1173): the node types of the grid as one 24-bit BMP per z-slice,
// <base>/<k>.bmp, biHeight = dimx, biWidth = dimy, x running from the top of the image down; NODE_IN blue (245, 73, 69 as B, G, R),
// NODE_OUT black, NODE_BOUND white, NODE_VALVE purple (241, 41, 212).  The reference pads a row with 3 * (dimy % 4) bytes; rows
// here are padded to a multiple of 4 bytes as the format asks (the same thing whenever dimy is a multiple of 4: every aligned grid).
#pragma once
#include <sys/stat.h>

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "AdiSolver3D_hip.h"

namespace fs3d {

template <typename FTYPE>
void OutputGridImages(const Grid3D<FTYPE> &g, const std::string &base)
{
    ::mkdir(base.c_str(), 0777);
    const int rowbytes = (3 * g.dimy + 3) / 4 * 4;
    unsigned char hdr[54] = {0};
    auto put32 = [&](int at, uint32_t v) { for (int b = 0; b < 4; b++) hdr[at + b] = (unsigned char)(v >> (8 * b)); };
    hdr[0] = 'B'; hdr[1] = 'M';
    put32(2, 54u + (uint32_t)rowbytes * g.dimx); put32(10, 54); put32(14, 40);
    put32(18, (uint32_t)g.dimy); put32(22, (uint32_t)g.dimx);
    hdr[26] = 1; hdr[28] = 24;                                   // planes, bits per pixel
    put32(46, 8);                                                // biClrUsed, as the reference sets it
    static const unsigned char col[4][3] = {{245, 73, 69}, {0, 0, 0}, {255, 255, 255}, {241, 41, 212}};
    std::vector<unsigned char> row((size_t)rowbytes, 0);
    for (int k = 0; k < g.dimz; k++) {
        const std::string name = base + "/" + std::to_string(k) + ".bmp";
        FILE *f = std::fopen(name.c_str(), "wb");
        if (!f) throw std::runtime_error("cannot create " + name);
        std::fwrite(hdr, 1, sizeof hdr, f);
        for (int i = g.dimx - 1; i >= 0; i--) {
            for (int j = 0; j < g.dimy; j++) {
                const unsigned char *c = col[g.type[g.Index(i, j, k)] & 3];
                row[3 * j] = c[0]; row[3 * j + 1] = c[1]; row[3 * j + 2] = c[2];
            }
            std::fwrite(row.data(), 1, row.size(), f);
        }
        std::fclose(f);
    }
}

}  // namespace fs3d
