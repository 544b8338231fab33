// Prints the LDS position (in float4 units) of every node of P2G's per-wave accumulation tile as csrc/layout.h TileSwz
// places it, for the dimension the file is compiled for (-DWGS_DIM): one line "x y z index". Host code only (no HIP call):
// tests/test_lds_layout.py reads the table and checks the bank-conflict claims of layout.h against the LDS rules of
// MI355X_MICROARCH.md.
#include <cstdio>
#include "layout.h"
int main() {
    using S = wgs::TileSwz<WGS_DIM>;
    constexpr int TW = wgs::Dim<WGS_DIM>::TW;
    std::printf("%d %d %d %d\n", WGS_DIM, TW, wgs::Dim<WGS_DIM>::BW, S::SIZE);
    for (int z = 0; z < (WGS_DIM == 3 ? TW : 1); z++)
        for (int y = 0; y < TW; y++)
            for (int x = 0; x < TW; x++) std::printf("%d %d %d %d %d\n", x, y, z, S::fx(x) + S::fy(y) + S::fz(z), S::of_tile(x + TW * y + TW * TW * z));
    return 0;
}
