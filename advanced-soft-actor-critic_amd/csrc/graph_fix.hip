// Fix-up pass over a captured hipGraph (the train step of `SAC_Base.train`, reference algorithm/sac_base.py:2494-2590, which
// this framework replays as one graph): every 1-D memset node becomes a kernel node running `k_graph_fill`.
//
// Why: on this ROCm (7.2, gfx950) a captured hipMemsetAsync of >= 16 bytes writes its value on the FIRST launch of the
// instantiated graph only — later launches fill the range with a stale 8-byte pattern (tools/debug/graph_memset_repro.py).
// ATen zeroes the semaphores of its split reductions with exactly such a memset (ATen/native/cuda/Reduce.cuh), so a captured
// `x.sum(0)` over thousands of rows — every nn.Linear's bias gradient — is wrong from the second replay on
// (tools/debug/graph_sum_repro.py).  Kernel nodes replay correctly; the pass swaps the node, keeps its edges.
#include "asac_common.h"

#include <vector>

namespace asac {
namespace gfix {

// dst[0 .. nbytes) <- the 32-bit pattern `pat` repeated from dst[0] (pattern byte k & 3 at byte k)
__global__ void __launch_bounds__(256) k_graph_fill(uint8_t* dst, uint32_t pat, uint64_t nbytes) {
    const uint64_t head = min((uint64_t)((16 - (reinterpret_cast<uintptr_t>(dst) & 15)) & 15), nbytes);
    const uint64_t body = (nbytes - head) >> 4;       // 16-byte stores
    const uint64_t tail0 = head + (body << 4);
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (uint64_t)gridDim.x * blockDim.x;
    if (tid < head) dst[tid] = (uint8_t)(pat >> (8 * (tid & 3)));
    if (tid < nbytes - tail0) dst[tail0 + tid] = (uint8_t)(pat >> (8 * ((tail0 + tid) & 3)));
    const int rot = 8 * (int)(head & 3);              // the word at byte offset head + 16 i starts at pattern byte head & 3
    const uint32_t w = rot ? (pat >> rot) | (pat << (32 - rot)) : pat;
    const uint4 v = make_uint4(w, w, w, w);
    uint4* d4 = reinterpret_cast<uint4*>(dst + head);
    for (uint64_t i = tid; i < body; i += nthreads) d4[i] = v;
}

inline bool ok(hipError_t e, const char* where) {
    if (e == hipSuccess) return true;
    set_error(e, where);
    return false;
}

}  // namespace gfix
}  // namespace asac

using namespace asac;
using namespace asac::gfix;

extern "C" {

int asac_graph_replace_memset_nodes(void* graph_, int* n_replaced, int* n_kept) {
    if (!graph_) return bad_arg("asac_graph_replace_memset_nodes");
    hipGraph_t graph = reinterpret_cast<hipGraph_t>(graph_);
    int replaced = 0, kept = 0;
    size_t n = 0;
    if (!ok(hipGraphGetNodes(graph, nullptr, &n), "hipGraphGetNodes")) return 1;
    std::vector<hipGraphNode_t> nodes(n);
    if (n && !ok(hipGraphGetNodes(graph, nodes.data(), &n), "hipGraphGetNodes")) return 1;
    for (hipGraphNode_t node : nodes) {
        hipGraphNodeType type;
        if (!ok(hipGraphNodeGetType(node, &type), "hipGraphNodeGetType")) return 1;
        if (type != hipGraphNodeTypeMemset) continue;
        hipMemsetParams mp{};
        if (!ok(hipGraphMemsetNodeGetParams(node, &mp), "hipGraphMemsetNodeGetParams")) return 1;
        const unsigned es = mp.elementSize;
        if (mp.height > 1 || !(es == 1 || es == 2 || es == 4) || !mp.dst || mp.width == 0) {
            ++kept;       // 2-D memsets: none in a train step so far; left as they are and reported
            continue;
        }
        uint32_t pat = mp.value;
        if (es == 1) pat = (pat & 0xffu) * 0x01010101u;
        else if (es == 2) pat = (pat & 0xffffu) * 0x00010001u;
        uint8_t* dst = static_cast<uint8_t*>(mp.dst);
        uint64_t nbytes = (uint64_t)mp.width * es;
        size_t n_in = 0, n_out = 0;
        if (!ok(hipGraphNodeGetDependencies(node, nullptr, &n_in), "hipGraphNodeGetDependencies")) return 1;
        if (!ok(hipGraphNodeGetDependentNodes(node, nullptr, &n_out), "hipGraphNodeGetDependentNodes")) return 1;
        std::vector<hipGraphNode_t> in(n_in), out(n_out);
        if (n_in && !ok(hipGraphNodeGetDependencies(node, in.data(), &n_in), "hipGraphNodeGetDependencies")) return 1;
        if (n_out && !ok(hipGraphNodeGetDependentNodes(node, out.data(), &n_out), "hipGraphNodeGetDependentNodes")) return 1;
        void* args[3] = {&dst, &pat, &nbytes};
        hipKernelNodeParams kp{};
        kp.func = reinterpret_cast<void*>(k_graph_fill);
        const uint64_t blocks = (nbytes / 16 + 255) / 256;
        kp.gridDim = dim3((unsigned)(blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks)));
        kp.blockDim = dim3(256);
        kp.sharedMemBytes = 0;
        kp.kernelParams = args;
        kp.extra = nullptr;
        hipGraphNode_t fill;
        if (!ok(hipGraphAddKernelNode(&fill, graph, in.data(), n_in, &kp), "hipGraphAddKernelNode")) return 1;
        if (n_out) {
            std::vector<hipGraphNode_t> from(n_out, fill);
            if (!ok(hipGraphAddDependencies(graph, from.data(), out.data(), n_out), "hipGraphAddDependencies")) return 1;
        }
        if (!ok(hipGraphDestroyNode(node), "hipGraphDestroyNode")) return 1;
        ++replaced;
    }
    if (n_replaced) *n_replaced = replaced;
    if (n_kept) *n_kept = kept;
    return 0;
}

}  // extern "C"
