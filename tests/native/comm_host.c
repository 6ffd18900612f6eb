/* A plain C host of the slab-sharded path ("Multi-GPU"): no Python, no torch.
 *
 *   comm_host                      one rank, self-sends: the sharded calls against the single-engine calls, bit for bit
 *   comm_host W R IDFILE [GPU]     rank R of W (one process per GPU; rank 0 writes the communicator id to IDFILE, the others
 *                                  wait for it): every rank reconstructs its slab, rank 0 prints the global scalars
 *
 * What it stands in for: the MPI host of tomofusion/cpu/utils/mpi_ctvlib.cpp (:400-422 ring exchange, :455,:547 all-reduce) and
 * the thread-per-GPU host of tomofusion/gpu/utils/multigpuengine.cpp:140-193. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "../../include/tomo_hip.h"

#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, tomo_last_error()); return 1; } } while (0)

static void fill(float *v, size_t n, unsigned seed)
{
    for (size_t i = 0; i < n; ++i) { seed = seed * 1664525u + 1013904223u; v[i] = (float)(seed >> 8) / 16777216.0f; }
}

int main(int argc, char **argv)
{
    const int world = argc > 3 ? atoi(argv[1]) : 1, rank = argc > 3 ? atoi(argv[2]) : 0, gpu = argc > 4 ? atoi(argv[4]) : rank;
    const int nglobal = 96, n = 48, np = 7;
    const int base = nglobal / world, rem = nglobal % world;
    const int nloc = base + (rank < rem ? 1 : 0), first = rank * base + (rank < rem ? rank : rem);
    double ang[7];
    for (int i = 0; i < np; ++i) ang[i] = (-60.0 + 20.0 * i) * M_PI / 180.0;
    size_t nvol = (size_t)nglobal * n * n;
    float *x = (float *)malloc(nvol * sizeof(float)), *out = (float *)malloc(nvol * sizeof(float)), *ref = (float *)malloc(nvol * sizeof(float));
    fill(x, nvol, 12345u);

    unsigned char id[128];
    if (rank == 0) {
        CHECK(tomo_comm_unique_id(id));
        if (argc > 3) { FILE *f = fopen(argv[3], "wb"); fwrite(id, 1, 128, f); fclose(f); }
    } else {
        FILE *f = NULL;
        for (int tries = 0; tries < 600 && !f; ++tries) { f = fopen(argv[3], "rb"); if (!f) usleep(100000); }
        if (!f || fread(id, 1, 128, f) != 128) { fprintf(stderr, "no communicator id\n"); return 1; }
        fclose(f);
    }
    tomo_engine *h = NULL;
    CHECK(tomo_create(nloc, n, np, ang, gpu, &h));
    CHECK(tomo_comm_init(h, id, world, rank));
    int w = 0, r = -1;
    CHECK(tomo_comm_info(h, &w, &r));
    if (w != world || r != rank) { fprintf(stderr, "comm_info %d %d\n", w, r); return 1; }
    const float *xloc = x + (size_t)first * n * n;
    CHECK(tomo_set_volume(h, TOMO_VOL_ORIGINAL, xloc));
    CHECK(tomo_forward_projection(h, TOMO_VOL_ORIGINAL, TOMO_SINO_B));          /* synthetic tilt series of the slab */
    CHECK(tomo_set_volume(h, TOMO_VOL_RECON, xloc));
    CHECK(tomo_copy_volume(h, TOMO_VOL_TEMP, TOMO_VOL_RECON));
    CHECK(tomo_sart_tracked(h, TOMO_VOL_RECON, TOMO_SINO_B, 0.5f, 1, NULL, TOMO_VOL_TEMP, TOMO_S_DIFF2));
    CHECK(tomo_data_distance_sq_async(h, TOMO_VOL_TEMP));
    CHECK(tomo_comm_tv_gd(h, 4, 0.3f, 1e-6f, TOMO_VOL_TEMP, TOMO_S_DIFF));
    double s[TOMO_S_COUNT];
    CHECK(tomo_comm_read_scalars(h, s, TOMO_S_COUNT));
    CHECK(tomo_get_volume(h, TOMO_VOL_RECON, out));
    if (rank == 0) printf("rank 0 of %d: dd^2 %.9e tv %.9e |step|^2 %.9e\n", world, s[TOMO_S_DD], s[TOMO_S_TV], s[TOMO_S_DIFF]);

    if (rank == 0) {        /* the same through the single-slab calls on ONE engine holding the whole volume (tomo_tv_gd_tracked wraps its own
                               halo planes): bit for bit at world 1; at world > 1 rank 0's slab and the global scalars to 2e-6 (the kernels pick
                               their vector width from the slab's slice count) */
        tomo_engine *one = NULL;
        CHECK(tomo_create(nglobal, n, np, ang, gpu, &one));
        CHECK(tomo_set_volume(one, TOMO_VOL_ORIGINAL, x));
        CHECK(tomo_forward_projection(one, TOMO_VOL_ORIGINAL, TOMO_SINO_B));
        CHECK(tomo_set_volume(one, TOMO_VOL_RECON, x));
        CHECK(tomo_copy_volume(one, TOMO_VOL_TEMP, TOMO_VOL_RECON));
        CHECK(tomo_sart_tracked(one, TOMO_VOL_RECON, TOMO_SINO_B, 0.5f, 1, NULL, TOMO_VOL_TEMP, TOMO_S_DIFF2));
        CHECK(tomo_data_distance_sq_async(one, TOMO_VOL_TEMP));
        CHECK(tomo_tv_gd_tracked(one, 4, 0.3f, 1e-6f, TOMO_VOL_TEMP, TOMO_S_DIFF));
        double s1[TOMO_S_COUNT];
        CHECK(tomo_read_scalars(one, s1, TOMO_S_COUNT));
        CHECK(tomo_get_volume(one, TOMO_VOL_RECON, ref));
        const float *refloc = ref + (size_t)first * n * n;
        const size_t nl = (size_t)nloc * n * n;
        if (world == 1) {
            if (memcmp(out, refloc, nl * sizeof(float)) != 0) { fprintf(stderr, "sharded (world 1) and single-slab volumes differ\n"); return 1; }
        } else {
            double num = 0, den = 0;
            for (size_t i = 0; i < nl; ++i) { double d = (double)out[i] - refloc[i]; num += d * d; den += (double)refloc[i] * refloc[i]; }
            if (sqrt(num / den) > 2e-6) { fprintf(stderr, "rank 0's slab is %.3e from the single engine\n", sqrt(num / den)); return 1; }
        }
        const int slots[4] = {TOMO_S_DD, TOMO_S_TV, TOMO_S_DIFF, TOMO_S_DIFF2};
        const double tol = world == 1 ? 1e-12 : 2e-6;
        for (int k = 0; k < 4; ++k)
            if (fabs(s[slots[k]] - s1[slots[k]]) > tol * fabs(s1[slots[k]])) { fprintf(stderr, "scalar %d: %.17g vs %.17g\n", slots[k], s[slots[k]], s1[slots[k]]); return 1; }
        CHECK(tomo_destroy(one));
    }
    {   /* the halo exchange and the TV value as bare calls (collective: every rank) */
        double sa[TOMO_S_COUNT];
        CHECK(tomo_comm_exchange_halo(h, TOMO_VOL_RECON));
        CHECK(tomo_tv_partial(h, TOMO_VOL_RECON, 1e-6f));
        CHECK(tomo_comm_read_scalars(h, sa, TOMO_S_COUNT));
        if (!(sa[TOMO_S_TV] > 0)) { fprintf(stderr, "tv %g\n", sa[TOMO_S_TV]); return 1; }
    }
    CHECK(tomo_comm_destroy(h));
    CHECK(tomo_destroy(h));
    free(x); free(out); free(ref);
    if (rank == 0) printf("COMM_HOST_OK\n");
    return 0;
}
