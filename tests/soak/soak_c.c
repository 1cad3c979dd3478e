/* soak_c.c — the C ABI alone (no Python, no torch) under a long run of small calls: fs_get_frontier_costs over 50 frontiers on a
 * 256 x 256 costmap, N times, resident set size from /proc/self/statm at the quarter points.  By hand on the GPU box:
 *   gcc -O2 -std=c99 -I include tests/soak/soak_c.c -o /tmp/soak_c -L fit-slam_amd/csrc -lfitslam_frontier -Wl,-rpath,$PWD/fit-slam_amd/csrc -lm && /tmp/soak_c 400000
 * A leak inside the library (or the HIP runtime under it) shows here; growth that only the Python soak shows belongs to the binding. */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "fitslam_frontier.h"

static double rss_mb(void)
{
    long pages = 0, resident = 0;
    FILE *f = fopen("/proc/self/statm", "r");
    if (!f) return -1.0;
    if (fscanf(f, "%ld %ld", &pages, &resident) != 2) resident = 0;
    fclose(f);
    return (double)resident * (double)sysconf(_SC_PAGESIZE) / 1048576.0;
}

#define CHECK(call) do { int rc_ = (call); if (rc_ != FS_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, fs_last_error(c)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const long total = argc > 1 ? atol(argv[1]) : 200000;
    const int with_fim = argc > 2 ? atoi(argv[2]) : 0;
    enum { NX = 256, N = 50, M = 20000 };
    fs_ctx *c = NULL;
    if (fs_ctx_create(0, NULL, &c) != FS_OK) { fprintf(stderr, "no gfx950 device\n"); return 2; }
    static unsigned char cells[NX * NX];
    unsigned s = 12345u;
    for (int i = 0; i < NX * NX; ++i) { s = s * 1664525u + 1013904223u; cells[i] = (s >> 24) < 120 ? 255 : ((s >> 24) < 235 ? 0 : 254); }
    fs_ray_params rp;
    memset(&rp, 0, sizeof rp);
    rp.max_camera_depth = 2.0; rp.delta_theta = 0.10; rp.camera_fov = 1.04; rp.robot_radius = 0.2; rp.n_rays = 0; rp.n_elev = 1;
    rp.obst_min = 240; rp.obst_max = 254; rp.trace_min = 255; rp.trace_max = 255; rp.factor_max = 1.2; rp.factor_min = 0.70;
    rp.polygon[0] = rp.polygon[1] = -1e9; rp.polygon[2] = rp.polygon[3] = 1e9;
    CHECK(fs_set_ray_params(c, &rp));
    const double origin[3] = {-6.4, -6.4, 0.0};
    CHECK(fs_upload_grid(c, cells, NX, NX, 1, origin, 0.05));
    CHECK(fs_set_arrival_limits(c, 4000.0, 10.0));
    static float lm[3 * M];
    for (int i = 0; i < 3 * M; ++i) { s = s * 1664525u + 1013904223u; lm[i] = ((float)(s >> 8) / 16777216.0f - 0.5f) * (i % 3 == 2 ? 2.0f : 12.0f); }
    if (with_fim) {
        fs_fim_params fp; fp.max_dist = 14.0; fp.max_angle = 4.0;
        CHECK(fs_upload_landmarks(c, lm, M));
        CHECK(fs_lookup_generate(c, NULL));
        CHECK(fs_set_fim_params(c, &fp));
    }
    double goal[3 * N], plen[N], phead[N], cost[N], au[N], du[N], pose[7] = {0, 0, 0, 0, 0, 0, 1};
    int32_t fsize[N], order[N];
    fs_record rec[N];
    for (int i = 0; i < N; ++i) {
        s = s * 1664525u + 1013904223u; goal[3 * i] = ((double)(s >> 8) / 16777216.0 - 0.5) * 10.0;
        s = s * 1664525u + 1013904223u; goal[3 * i + 1] = ((double)(s >> 8) / 16777216.0 - 0.5) * 10.0;
        goal[3 * i + 2] = 0.0; fsize[i] = 5; plen[i] = 1.0 + i * 0.3; phead[i] = fmod(i * 0.37, 3.0);
    }
    double mark[5] = {0};
    for (long k = 0; k < total; ++k) {
        if (k % (total / 4) == 0) mark[k / (total / 4)] = rss_mb();
        CHECK(fs_get_frontier_costs(c, N, goal, fsize, NULL, NULL, plen, phead, 0.25, 1.0, 0.5, 0.5, with_fim, rec, cost, au, du, order));
        if (with_fim) { float info; CHECK(fs_score_fim(c, 1, pose, &info, NULL, NULL, NULL, NULL, NULL)); }
    }
    mark[4] = rss_mb();
    printf("{\"calls\": %ld, \"with_fim\": %d, \"rss_mb_at_quarters\": [%.2f, %.2f, %.2f, %.2f, %.2f], \"growth_second_half_mb\": %.2f, \"first_cost\": %.6f}\n",
           total, with_fim, mark[0], mark[1], mark[2], mark[3], mark[4], mark[4] - mark[2], cost[order[0]]);
    fs_ctx_destroy(c);
    return 0;
}
