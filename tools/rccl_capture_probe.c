/* Probe (development tool): does the fork / join form of the library's RCCL exchange (csrc/exchange.hip: event record on the
 * compute stream, the grouped ncclSend / ncclRecv on a communication stream, event record, stream wait) record into a HIP-graph
 * capture of the compute stream and replay?  One GPU, a one-rank communicator, the exchange in loopback mode; every step
 * prints before it runs, so that a crash is attributable.  No torch in the process.
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tools/rccl_capture_probe.c -Lwxfactory_amd/lib -lwxhip \
 *       -L/opt/rocm/lib -lamdhip64 -o rccl_capture_probe
 *   ./rccl_capture_probe [global|threadlocal|relaxed] [inline] [multi] [prio] [ec=N] */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "wxhip.h"

#define SAY(...) do { printf(__VA_ARGS__); printf("\n"); fflush(stdout); } while (0)
#define CHECK_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { SAY("%s: %s", #call, hipGetErrorString(e_)); exit(2); } } while (0)
#define CHECK_WX(call) do { if ((call) != WX_OK) { SAY("%s: %s", #call, wx_last_error()); exit(3); } } while (0)

static size_t ec = 4096;
static wx_exchange* ex;
static hipStream_t compute, comms;
static double *pattern, *scratch;
static size_t nsend;
static int inline_form, multi, prio;

/* one "evaluation" on the compute stream: pack (pattern -> send slots), start, interior work beside the exchange, wait,
 * boundary work (halos gathered in (tile, edge) order into scratch) */
static void enqueue(void) {
    int t, e;
    for (t = 0; t < 6; ++t)
        for (e = 0; e < 4; ++e)
            CHECK_HIP(hipMemcpyAsync(wx_exchange_send_ptr(ex, t, e), pattern + ((size_t)t * 4 + e) * ec, ec * sizeof(double),
                                     hipMemcpyDeviceToDevice, compute));
    SAY("  start");
    CHECK_WX(wx_exchange_start(ex, compute, inline_form ? compute : comms));
    CHECK_HIP(hipMemsetAsync(scratch, 0, nsend * sizeof(double), compute));
    SAY("  wait");
    CHECK_WX(wx_exchange_wait(ex, compute));
    for (t = 0; t < 6; ++t)
        for (e = 0; e < 4; ++e)
            CHECK_HIP(hipMemcpyAsync(scratch + ((size_t)t * 4 + e) * ec, wx_exchange_halo_ptr(ex, t, e), ec * sizeof(double),
                                     hipMemcpyDeviceToDevice, compute));
}

int main(int argc, char** argv) {
    hipStreamCaptureMode mode = hipStreamCaptureModeGlobal;
    int a, t, e, rep;
    unsigned char id[WX_COMM_ID_BYTES];
    wx_comm* comm = NULL;
    hipGraph_t graph = NULL;
    hipGraphExec_t exec = NULL;
    double* host;
    size_t i;
    for (a = 1; a < argc; ++a) {
        if (!strcmp(argv[a], "threadlocal")) mode = hipStreamCaptureModeThreadLocal;
        if (!strcmp(argv[a], "relaxed")) mode = hipStreamCaptureModeRelaxed;
        if (!strcmp(argv[a], "inline")) inline_form = 1;
        if (!strcmp(argv[a], "multi")) multi = 1;
        if (!strcmp(argv[a], "prio")) prio = 1;            /* streams as torch's pool makes them: hipStreamCreateWithPriority */
        if (!strncmp(argv[a], "ec=", 3)) ec = (size_t)atol(argv[a] + 3);   /* doubles per edge message */
    }
    SAY("%s, RCCL %d, capture mode %d, %s form", wx_version(), wx_comm_rccl_version(), (int)mode,
        inline_form ? "stream-ordered" : "fork/join");
    CHECK_WX(wx_comm_unique_id(id));
    SAY("comm init");
    CHECK_WX(wx_comm_init_rank(&comm, 1, id, 0));
    CHECK_WX(wx_exchange_create(&ex, comm, 0, 1, 1, ec, 1));
    CHECK_WX(wx_exchange_bind(ex, NULL, NULL));
    nsend = wx_exchange_send_doubles(ex);
    if (prio) {
        int lo = 0, hi = 0;
        CHECK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        SAY("priority streams (range %d .. %d)", lo, hi);
        CHECK_HIP(hipStreamCreateWithPriority(&compute, hipStreamNonBlocking, lo));
        CHECK_HIP(hipStreamCreateWithPriority(&comms, hipStreamNonBlocking, lo));
    } else {
        CHECK_HIP(hipStreamCreateWithFlags(&compute, hipStreamNonBlocking));
        CHECK_HIP(hipStreamCreateWithFlags(&comms, hipStreamNonBlocking));
    }
    host = (double*)malloc(nsend * sizeof(double));
    CHECK_HIP(hipMalloc((void**)&pattern, nsend * sizeof(double)));
    CHECK_HIP(hipMalloc((void**)&scratch, nsend * sizeof(double)));

    for (rep = 0; rep < 4; ++rep) {
        int bad = 0;
        /* what tile t sends through edge e: the value 1000 rep + 10 t + e in every word */
        for (i = 0; i < nsend; ++i) host[i] = 1000.0 * rep + 10.0 * (double)((i / ec) / 4) + (double)((i / ec) % 4);
        CHECK_HIP(hipMemcpy(pattern, host, nsend * sizeof(double), hipMemcpyHostToDevice));
        if (rep == 0) {
            SAY("eager");
            enqueue();
        } else {
            if (rep == 1) {
                SAY("begin capture");
                CHECK_HIP(hipStreamBeginCapture(compute, mode));
                enqueue();
                SAY("end capture");
                CHECK_HIP(hipStreamEndCapture(compute, &graph));
                SAY("instantiate");
                CHECK_HIP(hipGraphInstantiate(&exec, graph, NULL, NULL, 0));
            }
            SAY("replay %d", rep);
            CHECK_HIP(hipGraphLaunch(exec, compute));
        }
        CHECK_HIP(hipStreamSynchronize(compute));
        CHECK_HIP(hipMemcpy(host, scratch, nsend * sizeof(double), hipMemcpyDeviceToHost));
        for (t = 0; t < 6; ++t)
            for (e = 0; e < 4; ++e) {
                int qt, qe;
                CHECK_WX(wx_exchange_neighbor(ex, t, e, &qt, &qe, NULL));
                /* edge e of tile t holds what tile qt sent through its edge qe */
                for (i = 0; i < ec; ++i)
                    if (host[((size_t)t * 4 + e) * ec + i] != 1000.0 * rep + 10.0 * qt + qe) bad = 1;
            }
        SAY("%s %d: halos %s", rep == 0 ? "eager" : "replay", rep, bad ? "WRONG" : "correct");
        if (bad) return 5;
    }
    if (multi) {
        /* what a torch process does over a test session: more captures, each from a NEW origin stream (torch.cuda.graph is
         * given a fresh side stream), the communication stream alternating between two (one per exchange object), an eager
         * exchange on yet another pair of streams in between (the same communicator throughout) */
        hipStream_t comm2, eager_c, eager_m;
        int round;
        CHECK_HIP(hipStreamCreateWithFlags(&comm2, hipStreamNonBlocking));
        CHECK_HIP(hipStreamCreateWithFlags(&eager_c, hipStreamNonBlocking));
        CHECK_HIP(hipStreamCreateWithFlags(&eager_m, hipStreamNonBlocking));
        for (round = 0; round < 4; ++round) {
            hipGraph_t g2 = NULL;
            hipGraphExec_t x2 = NULL;
            hipStream_t keep_c = compute, keep_m = comms;
            SAY("round %d: eager exchange on a third pair of streams", round);
            compute = eager_c; comms = eager_m;
            enqueue();
            CHECK_HIP(hipStreamSynchronize(compute));
            SAY("round %d: capture from a new origin stream, communication stream %d", round, round & 1);
            CHECK_HIP(hipStreamCreateWithFlags(&compute, hipStreamNonBlocking));
            comms = (round & 1) ? comm2 : keep_m;
            CHECK_HIP(hipStreamBeginCapture(compute, mode));
            enqueue();
            SAY("  end capture");
            CHECK_HIP(hipStreamEndCapture(compute, &g2));
            CHECK_HIP(hipGraphInstantiate(&x2, g2, NULL, NULL, 0));
            CHECK_HIP(hipGraphLaunch(x2, compute));
            CHECK_HIP(hipStreamSynchronize(compute));
            SAY("  replayed");
            CHECK_HIP(hipGraphExecDestroy(x2));
            CHECK_HIP(hipGraphDestroy(g2));
            CHECK_HIP(hipStreamDestroy(compute));
            compute = keep_c; comms = keep_m;
        }
    }
    SAY("destroy graph, exchange, communicator");
    CHECK_HIP(hipGraphExecDestroy(exec));
    CHECK_HIP(hipGraphDestroy(graph));
    CHECK_WX(wx_exchange_destroy(ex));
    CHECK_WX(wx_comm_destroy(comm));
    SAY("PASS");
    return 0;
}
