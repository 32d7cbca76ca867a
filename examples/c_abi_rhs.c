/* A plain-C caller of libwxhip.so: no Python, no torch - device memory from the HIP runtime, the plan, the two launches of
 * one R(Q) on one cubed-sphere tile, the reference's nine-stamp timing row (rhs/rhs.py:88-118) and a comparison with the
 * expected R.  Inputs are raw little-endian float64 files in one directory (written by tests/test_c_abi_example_gpu.py
 * from a golden fixture, in the layouts include/wxhip.h documents):
 *     meta.txt                     "n H V case_number panel"
 *     ops_{extrap_neg,extrap_pos,diff_solpt,correction,highfilter}.bin
 *     metric_<field>.bin           every member of wx_euler3d_metric (damp_* only for cases 21 / 22)
 *     q.bin, halo_{0,1,2,3}.bin    the state (5,V,H,H,n^3) and the four received edge messages (5,V,H,n^2)
 *     r.bin                        the expected right-hand side
 * Build:  gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_abi_rhs.c \
 *             -Lwxfactory_amd/lib -lwxhip -L/opt/rocm/lib -lamdhip64 -lm -o c_abi_rhs
 * Exit code 0 when max |R - R_expected| / max |R_expected| < 1e-10 for every variable. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "wxhip.h"

#define CHECK_HIP(call)                                                                    \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); exit(2); } \
    } while (0)
#define CHECK_WX(call)                                                                     \
    do {                                                                                   \
        if ((call) != WX_OK) { fprintf(stderr, "%s: %s\n", #call, wx_last_error()); exit(3); }        \
    } while (0)

static double* read_file(const char* dir, const char* name, size_t count, int optional) {
    char path[1024];
    snprintf(path, sizeof(path), "%s/%s", dir, name);
    FILE* f = fopen(path, "rb");
    if (!f) {
        if (optional) return NULL;
        fprintf(stderr, "cannot open %s\n", path);
        exit(4);
    }
    double* a = (double*)malloc(count * sizeof(double));
    if (!a || fread(a, sizeof(double), count, f) != count) { fprintf(stderr, "short read of %s (%zu doubles)\n", path, count); exit(4); }
    fclose(f);
    return a;
}

static double* to_device(const double* host, size_t count) {
    double* d = NULL;
    if (!host) return NULL;
    CHECK_HIP(hipMalloc((void**)&d, count * sizeof(double)));
    CHECK_HIP(hipMemcpy(d, host, count * sizeof(double), hipMemcpyHostToDevice));
    return d;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s <directory>\n", argv[0]); return 1; }
    const char* dir = argv[1];
    int n, H, V, case_number, panel;
    {
        char path[1024];
        snprintf(path, sizeof(path), "%s/meta.txt", dir);
        FILE* f = fopen(path, "r");
        if (!f || fscanf(f, "%d %d %d %d %d", &n, &H, &V, &case_number, &panel) != 5) { fprintf(stderr, "bad meta.txt\n"); return 4; }
        fclose(f);
    }
    const size_t n2 = (size_t)n * n, n3 = n2 * n, elems = (size_t)V * H * H, pts = elems * n3;
    printf("%s, %d devices; tile n=%d H=%d V=%d case %d panel %d\n", wx_version(), wx_device_count(), n, H, V, case_number, panel);

    wx_dfr_ops ops;
    ops.extrap_neg = read_file(dir, "ops_extrap_neg.bin", n, 0);
    ops.extrap_pos = read_file(dir, "ops_extrap_pos.bin", n, 0);
    ops.diff_solpt = read_file(dir, "ops_diff_solpt.bin", n2, 0);
    ops.correction = read_file(dir, "ops_correction.bin", 2 * (size_t)n, 0);
    ops.highfilter = read_file(dir, "ops_highfilter.bin", n2, 0);

    wx_euler3d_metric m;
    memset(&m, 0, sizeof(m));
    const size_t nh = (size_t)H + 2, nv = (size_t)V + 2;
    m.sqrtG = to_device(read_file(dir, "metric_sqrtG.bin", pts, 0), pts);
    m.h_contra = to_device(read_file(dir, "metric_h_contra.bin", 9 * pts, 0), 9 * pts);
    m.christoffel = to_device(read_file(dir, "metric_christoffel.bin", 27 * pts, 0), 27 * pts);
    m.inv_dzdeta = to_device(read_file(dir, "metric_inv_dzdeta.bin", pts, 0), pts);
    m.sqrtG_itf_i = to_device(read_file(dir, "metric_sqrtG_itf_i.bin", (size_t)V * H * nh * 2 * n2, 0), (size_t)V * H * nh * 2 * n2);
    m.sqrtG_itf_j = to_device(read_file(dir, "metric_sqrtG_itf_j.bin", (size_t)V * nh * H * 2 * n2, 0), (size_t)V * nh * H * 2 * n2);
    m.sqrtG_itf_k = to_device(read_file(dir, "metric_sqrtG_itf_k.bin", nv * H * H * 2 * n2, 0), nv * H * H * 2 * n2);
    m.h_contra_itf_i = to_device(read_file(dir, "metric_h_contra_itf_i.bin", 9 * (size_t)V * H * nh * 2 * n2, 0), 9 * (size_t)V * H * nh * 2 * n2);
    m.h_contra_itf_j = to_device(read_file(dir, "metric_h_contra_itf_j.bin", 9 * (size_t)V * nh * H * 2 * n2, 0), 9 * (size_t)V * nh * H * 2 * n2);
    m.h_contra_itf_k = to_device(read_file(dir, "metric_h_contra_itf_k.bin", 9 * nv * H * H * 2 * n2, 0), 9 * nv * H * H * 2 * n2);
    m.damp_coef = to_device(read_file(dir, "metric_damp_coef.bin", pts, 1), pts);
    m.damp_uref = to_device(read_file(dir, "metric_damp_uref.bin", 3 * pts, 1), 3 * pts);
    m.boundary_sn = to_device(read_file(dir, "metric_boundary_sn.bin", (size_t)H * n, 0), (size_t)H * n);
    m.boundary_we = to_device(read_file(dir, "metric_boundary_we.bin", (size_t)H * n, 0), (size_t)H * n);

    wx_euler3d_plan* plan = NULL;
    CHECK_WX(wx_euler3d_plan_create(&plan, n, H, V, case_number, WX_F64, panel, &ops, &m));
    const size_t edge = wx_euler3d_edge_count(plan);
    printf("plan: %zu doubles per edge message, %.0f compulsory bytes per point, matrix cores: %d\n", edge,
           wx_euler3d_bytes_per_point(plan), wx_euler3d_uses_matrix_cores(plan, WX_KERNEL_RHS));

    double* q = to_device(read_file(dir, "q.bin", 5 * pts, 0), 5 * pts);
    double* r_expected = read_file(dir, "r.bin", 5 * pts, 0);
    double* r = NULL;
    CHECK_HIP(hipMalloc((void**)&r, 5 * pts * sizeof(double)));
    void *send[4], *halo[4];
    char name[32];
    for (int e = 0; e < 4; ++e) {
        CHECK_HIP(hipMalloc(&send[e], edge * sizeof(double)));
        snprintf(name, sizeof(name), "halo_%d.bin", e);
        halo[e] = to_device(read_file(dir, name, edge, 0), edge);
    }

    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    wx_phase_timer* timer = NULL;
    CHECK_WX(wx_phase_timer_create(&timer));
    CHECK_WX(wx_phase_timer_stamp(timer, 0, stream));
    CHECK_WX(wx_euler3d_extrap_pack(plan, q, send, stream));          /* phases 1-2: faces + edge messages */
    CHECK_WX(wx_phase_timer_stamp(timer, 1, stream));
    /* (a distributed caller posts its exchange of `send` here - INTEGRATION.md section 4 - and evaluates WX_REGION_INTERIOR
     * meanwhile; this example was handed the received messages) */
    CHECK_WX(wx_phase_timer_stamp(timer, 5, stream));
    CHECK_WX(wx_euler3d_rhs(plan, q, (const void* const*)halo, r, WX_REGION_ALL, stream));   /* phases 3-8 */
    CHECK_WX(wx_phase_timer_stamp(timer, 8, stream));
    double seconds[9];
    CHECK_WX(wx_phase_timer_elapsed(timer, seconds));
    printf("timing row [s]:");
    for (int i = 0; i < 9; ++i) printf(" %.3e", seconds[i]);
    printf("\n");

    double* r_host = (double*)malloc(5 * pts * sizeof(double));
    CHECK_HIP(hipMemcpy(r_host, r, 5 * pts * sizeof(double), hipMemcpyDeviceToHost));
    int ok = 1;
    for (int v = 0; v < 5; ++v) {
        double err = 0.0, scale = 0.0;
        for (size_t i = 0; i < pts; ++i) {
            const double d = fabs(r_host[v * pts + i] - r_expected[v * pts + i]);
            if (!(d <= err)) err = d;   /* (a NaN sticks) */
            if (fabs(r_expected[v * pts + i]) > scale) scale = fabs(r_expected[v * pts + i]);
        }
        printf("variable %d: max |R - R_expected| = %.3e, max |R_expected| = %.3e\n", v, err, scale);
        if (!(err <= 1e-10 * scale)) ok = 0;
    }
    /* The same evaluation with the halos TRAVELLING: a one-rank RCCL communicator, the library's exchange in loopback mode
     * (every edge message through ncclSend / ncclRecv on a communication stream forked from the compute stream by an
     * event), INTERIOR launch while it is in flight, join, BOUNDARY launch - the several-GPU path of
     * process_topology.py:269-386, 564-606 + rhs/rhs.py:88-118 from C, on one GPU.  This program holds one tile: the
     * messages its neighbours would have packed are the halo files, put where the neighbours' pack kernels would have
     * written them. */
    {
        unsigned char id[WX_COMM_ID_BYTES];
        wx_comm* comm = NULL;
        wx_exchange* ex = NULL;
        hipStream_t comm_stream;
        int e, tile = panel, same = 1;   /* one tile per panel: tile id = panel */
        size_t i;
        double* r2 = NULL;
        double* r2_host = (double*)malloc(5 * pts * sizeof(double));
        void* send2[4];
        const void* halo2[4];
        CHECK_WX(wx_comm_unique_id(id));
        CHECK_WX(wx_comm_init_rank(&comm, 1, id, 0));
        CHECK_WX(wx_exchange_create(&ex, comm, 0, 1, 1, edge, 1));
        CHECK_WX(wx_exchange_bind(ex, NULL, NULL));
        CHECK_HIP(hipStreamCreateWithFlags(&comm_stream, hipStreamNonBlocking));
        CHECK_HIP(hipMalloc((void**)&r2, 5 * pts * sizeof(double)));
        CHECK_HIP(hipMemset(r2, 0, 5 * pts * sizeof(double)));
        for (e = 0; e < 4; ++e) {
            int q_tile, q_edge, q_rank;
            CHECK_WX(wx_exchange_neighbor(ex, tile, e, &q_tile, &q_edge, &q_rank));
            CHECK_HIP(hipMemcpy(wx_exchange_send_ptr(ex, q_tile, q_edge), halo[e], edge * sizeof(double), hipMemcpyDeviceToDevice));
            send2[e] = wx_exchange_send_ptr(ex, tile, e);
            halo2[e] = wx_exchange_halo_ptr(ex, tile, e);
        }
        CHECK_WX(wx_euler3d_extrap_pack(plan, q, send2, stream));
        CHECK_WX(wx_exchange_start(ex, stream, comm_stream));
        CHECK_WX(wx_euler3d_rhs(plan, q, NULL, r2, WX_REGION_INTERIOR, stream));
        CHECK_WX(wx_exchange_wait(ex, stream));
        CHECK_WX(wx_euler3d_rhs(plan, q, halo2, r2, WX_REGION_BOUNDARY, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        CHECK_HIP(hipMemcpy(r2_host, r2, 5 * pts * sizeof(double), hipMemcpyDeviceToHost));
        for (i = 0; i < 5 * pts; ++i)
            if (r2_host[i] != r_host[i]) same = 0;
        printf("RCCL %d, loopback exchange + INTERIOR / BOUNDARY launches: %s\n", wx_comm_rccl_version(),
               same ? "bit-identical to the single launch" : "DIFFERENT");
        if (!same) ok = 0;
        /* a caller's reduction on the same communicator (solvers/global_operations.py:14-36: the Krylov solvers' dot products
         * and norms), in stream order; one rank: the sum of one contribution.  And the order of teardown: the communicator
         * refuses to go while an exchange made on it is alive. */
        {
            double dots[3] = {1.5, -2.0, 4.0}, back[3];
            double* ddots = NULL;
            CHECK_HIP(hipMalloc((void**)&ddots, sizeof dots));
            CHECK_HIP(hipMemcpy(ddots, dots, sizeof dots, hipMemcpyHostToDevice));
            CHECK_WX(wx_comm_allreduce(comm, ddots, 3, WX_REDUCE_SUM, stream));
            CHECK_HIP(hipStreamSynchronize(stream));
            CHECK_HIP(hipMemcpy(back, ddots, sizeof dots, hipMemcpyDeviceToHost));
            if (back[0] != 1.5 || back[1] != -2.0 || back[2] != 4.0) ok = 0;
            CHECK_HIP(hipFree(ddots));
            if (wx_comm_users(comm) != 1 || wx_comm_destroy(comm) == WX_OK) ok = 0;
            printf("wx_comm_allreduce on the exchange's communicator: %s; HIP runtime %d\n", ok ? "ok" : "WRONG", wx_hip_runtime_version());
        }
        CHECK_WX(wx_exchange_destroy(ex));
        CHECK_WX(wx_comm_destroy(comm));
        CHECK_HIP(hipStreamDestroy(comm_stream));
        free(r2_host);
    }

    /* an invalid call reports through the status and wx_last_error, as the header says */
    if (wx_euler3d_rhs(plan, q, NULL, r, WX_REGION_ALL, stream) == WX_OK) ok = 0;
    else printf("expected refusal: %s\n", wx_last_error());

    CHECK_WX(wx_phase_timer_destroy(timer));
    CHECK_WX(wx_euler3d_plan_destroy(plan));
    CHECK_HIP(hipStreamDestroy(stream));
    printf("%s\n", ok ? "PASS" : "FAIL");
    return ok ? 0 : 5;
}
