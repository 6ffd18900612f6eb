// resident.h -- host tables of the volume-RESIDENT SART sweep (kernel: k_sart_resident, sart_resident.hip.h; builder: resident.cpp).
//
// A 64-slice chunk of the whole image stays in the vector registers of the chip for all angles of a sweep: one workgroup of 16 waves
// per 32 x 32-pixel tile (one per CU), wave w holding the 8 x 8 block (w / 4, w % 4) of the tile as 64 registers of 64 lanes (lane =
// slice).  Per angle the workgroups exchange only ray sums: every wave forms the partial sums of the (at most 14) rays that cross its
// block, the workgroup adds them per ray of the tile's window, a fixed reducer (tile j / rpt) adds a ray's tile sums in ascending
// tile order and publishes the normalised residual row, and every tile picks up the rows of its window.
// What the kernel needs from the host, per angle i and tile k:
//   hdr[i*ntiles + k]   jbase = first ray of the tile's window, nrays = its length (<= MAXWIN), dw[w] = first ray of wave w's block
//                       window minus jbase
//   cell                per (i, k, wave, pixel q of the block, row-major) one 16-byte cell {slot0, w0, w1, 1/(w0+w1)}, read by SCALAR
//                       loads in both passes: slot0 = the pixel's first ray - first ray of the wave's window; its second ray is ALWAYS
//                       the next one (slot0 + 1: the neighbour property build_tables checks as art_chain_ok -- one register index per
//                       pixel picks both); a pixel no ray crosses: slot0 = SINK, weights 0, divisor 1
//   ts[(i*ntiles+k)*MAXWIN + r]  the block sums that make up the tile's sum of window ray r: up to TSN bytes wave << 4 | slot,
//                       ascending wave, padded with TS_PAD (a block-sum row that is always zero)
//   rl[(i*N + j)*RL ..] the tiles whose window holds ray j, ascending, as row ids tile * MAXWIN + (j - jbase); 0xFFFF ends the list
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "sysmat.h"

namespace tomo {

struct Resident {
    static constexpr int T = 32, B = 8, WAVES = 16, PPW = 64, NSLOT = 16, USABLE = 14, SINK = 14, MAXWIN = 48, RL = 32, TSN = 8, TS_PAD = 0x0E;
    struct Hdr { uint16_t jbase, nrays; uint8_t dw[WAVES]; uint8_t pad[12]; };
    static_assert(sizeof(Hdr) == 32, "header layout");
    bool ok = false;
    std::string why;
    int tiles = 0, ntiles = 0;      // tiles per image side, tiles of the image
    int rpt = 0;                    // rays of an angle a tile reduces: [tile * rpt, tile * rpt + rpt)
    std::vector<Hdr> hdr;           // [P * ntiles]
    std::vector<uint32_t> cell;     // [P * ntiles * WAVES * PPW * 4]
    std::vector<uint8_t> ts;        // [P * ntiles * MAXWIN * TSN]
    std::vector<uint16_t> rl;       // [P * N * RL]
    // pixel q of wave w inside its tile
    static void pixel(int w, int q, int &ly, int &lz) { ly = (w >> 2) * B + (q >> 3); lz = (w & 3) * B + (q & 7); }
};

// needs t.cell (build_tables).  max_tiles: workgroups that can be resident at once (CUs of the device).
void build_sart_resident(int N, int P, const Tables &t, int max_tiles, Resident &r);

}  // namespace tomo
