// Writes a small result file through host/NetCDF3.h (no GPU involved); tests/test_host_driver.py reads it back.
#include <cstdio>
#include <vector>
#include "../cmc_fluid_solver_amd/host/NetCDF3.h"
int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    const int nx = 3, ny = 4, nz = 5;
    const float bbox[6] = {-0.5f, 0.25f, 0.0f, 1.5f, 2.25f, 1.0f};
    fs3d::NetCDF3Writer nc;
    if (argc > 2) {                                        // sea mode: degree units and the depth variable `d` (x, y)
        std::vector<float> d(nx * ny);
        for (int c = 0; c < nx * ny; c++) d[c] = -10.0f * c + 0.5f;
        nc.Create(argv[1], bbox, 0.5, 10.0, nx, ny, nz, {"u", "d"}, true, d.data());
    } else
    nc.Create(argv[1], bbox, 0.5, 10.0, nx, ny, nz, {"u", "w", "T"});
    std::vector<float> vel(nx * ny * nz * 3);
    std::vector<double> T(nx * ny * nz);
    for (int layer = 0; layer < 3; layer++) {
        for (int c = 0; c < nx * ny * nz; c++) {
            vel[3 * c + 0] = 100.0f * layer + c; vel[3 * c + 1] = -1.0f; vel[3 * c + 2] = 0.5f * c - layer;
            T[c] = c == 7 ? 99999.0 : 1.0 + 0.001 * c + layer;
        }
        nc.AppendLayer(vel.data(), T.data());
    }
    std::printf("%u\n", nc.NumRecords());
    return 0;
}
