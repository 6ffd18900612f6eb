// sysmat.h -- host-side system matrix + derived tables (see sysmat.cpp).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace tomo {

struct Coo {  // CSR-shaped storage; "Coo" because rows may still be in path order before sort_rows()
    int64_t nrow = 0, ncol = 0;
    std::vector<int64_t> ptr;
    std::vector<uint32_t> col;
    std::vector<float> val;
};

struct Cell {  // the (at most two) rays of one angle that cross one pixel
    uint32_t r0; float w0; uint32_t r1; float w1;
};

struct Tables {
    std::vector<float> rowsum, rowinner, colsum_all;
    std::vector<Cell> cell;  // [P][N*N]
    float lipschitz = 0.f;
};

void build_parallel_ray(int N, int P, const double *angles_rad, Coo &out);
bool coo_from_triplets(int64_t nrow, int64_t ncol, int64_t nnz, const float *rows, const float *cols,
                       const float *vals, Coo &out, std::string &err);
void sort_rows(Coo &m);
bool build_tables(const Coo &m, int N, int P, Tables &t, std::string &err);

}  // namespace tomo
