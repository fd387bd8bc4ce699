// graph.hip — hipGraph capture / replay of a launch sequence (one denoise step: forward body, LM-head rows, select and
// commit kernels).  Every per-step quantity of the sampler is schedule-determined (text k, image mask_len,
// which forwards run), so a step is a fixed sequence of kernels over fixed buffers and can be replayed.
// Thin, allocation-free wrappers so that a host without torch (or without a tracing compiler) gets the same facility.
#include "../../include/mmada_mi355x.h"
#include "kernels.h"

struct mmada_graph {
    hipGraphExec_t exec = nullptr;
    int nodes = 0;
};

extern "C" {

int mmada_graph_begin(void* stream) {
    if (!stream) return mm_fail("mmada_graph_begin: the legacy default stream cannot be captured; pass a created stream");
    // thread-local: another host thread (a monitor, a loader) may keep calling HIP while this thread captures
    MM_CHECK_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
    return 0;
}

int mmada_graph_end(void* stream, mmada_graph** out) {
    if (!stream || !out) return mm_fail("mmada_graph_end: null argument");
    hipGraph_t g = nullptr;
    MM_CHECK_HIP(hipStreamEndCapture((hipStream_t)stream, &g));
    if (!g) return mm_fail("mmada_graph_end: the capture was invalidated (a call in the sequence is not capturable)");
    mmada_graph* r = new mmada_graph();
    size_t n = 0;
    if (hipGraphGetNodes(g, nullptr, &n) == hipSuccess) r->nodes = (int)n;
    hipError_t e = hipGraphInstantiate(&r->exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) {
        delete r;
        return mm_fail("mmada_graph_end: hipGraphInstantiate: %s", hipGetErrorString(e));
    }
    *out = r;
    return 0;
}

int mmada_graph_abort(void* stream) {
    // leave capture mode after a failed call inside the sequence; the partial graph is dropped
    hipGraph_t g = nullptr;
    (void)hipStreamEndCapture((hipStream_t)stream, &g);
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
    return 0;
}

int mmada_graph_launch(mmada_graph* g, void* stream) {
    if (!g || !g->exec) return mm_fail("mmada_graph_launch: null graph");
    MM_CHECK_HIP(hipGraphLaunch(g->exec, (hipStream_t)stream));
    return 0;
}

int mmada_graph_num_nodes(const mmada_graph* g) { return g ? g->nodes : 0; }

int mmada_graph_destroy(mmada_graph* g) {
    if (!g) return 0;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    delete g;
    return 0;
}

}  // extern "C"
