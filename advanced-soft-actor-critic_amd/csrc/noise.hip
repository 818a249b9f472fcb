// Every random draw of one train step in ONE launch (graph-capturable): the f64 uniforms of the
// stratified PER sample (reference replay_buffer.py:196, NumPy global RNG) and the Gaussian noise of
// every rsample of the step (reference `Normal.rsample`, sac_base.py:1346, 1883, 1927 — torch global
// RNG).  The reference's streams are not reproducible across frameworks anyway (parity tests inject
// recorded draws instead); what matters here is that replaying a captured graph produces FRESH draws:
// the Philox4x32-10 counter is (lane index, *step_counter) with the counter read from device memory,
// so the same frozen launch yields a new block of numbers after every train step.
#include "asac_common.h"
#include "asac_noise.h"

#include <cmath>

namespace asac {

__global__ __launch_bounds__(256) void k_noise_fill(const PrologueArgs a) { prologue_block(a, (int)blockIdx.x, false); }

}  // namespace asac


using namespace asac;

extern "C" {

int asac_noise_fill(uint64_t seed, const int64_t* step_counter, double* uniform_out, int64_t n_uniform,
                    float* normal_out, int64_t n_normal, int32_t* subsets_out, int n_subsets, int E_sample,
                    int E, void* stream) {
    if (!step_counter || n_uniform < 0 || n_normal < 0 || n_subsets < 0 || (n_uniform > 0 && !uniform_out) ||
        (n_normal > 0 && !normal_out) || n_uniform + n_normal + n_subsets == 0)
        return bad_arg("asac_noise_fill");
    if (n_subsets > 0 && (!subsets_out || E_sample < 1 || E_sample > E || E > ASAC_MAX_ENSEMBLE))
        return bad_arg("asac_noise_fill: subsets");
    const int64_t lanes = (n_normal + 3) / 4 + (n_uniform + 1) / 2 + n_subsets;
    const PrologueArgs a{seed, step_counter, uniform_out, n_uniform, normal_out, n_normal, subsets_out, n_subsets, E_sample, E,
                         0, nullptr, nullptr, 0, 0.f, 0.f, 0, nullptr, 0};
    ASAC_LAUNCH(k_noise_fill, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, as_stream(stream), a);
    return finish_launch("asac_noise_fill");
}

int asac_step_prologue(float* target, const float* source, int64_t n_polyak, float tau, float* zero_out,
                       int64_t n_zero, uint64_t seed, const int64_t* step_counter, double* uniform_out,
                       int64_t n_uniform, float* normal_out, int64_t n_normal, int32_t* subsets_out, int n_subsets,
                       int E_sample, int E, void* stream) {
    if (!step_counter || n_uniform < 0 || n_normal < 0 || n_subsets < 0 || n_polyak < 0 || n_zero < 0 ||
        n_polyak + n_zero == 0 || (n_polyak > 0 && (!target || !source)) || (n_zero > 0 && !zero_out) ||
        (n_uniform > 0 && !uniform_out) || (n_normal > 0 && !normal_out) || n_uniform + n_normal + n_subsets == 0)
        return bad_arg("asac_step_prologue");
    if (n_subsets > 0 && (!subsets_out || E_sample < 1 || E_sample > E || E > ASAC_MAX_ENSEMBLE))
        return bad_arg("asac_step_prologue: subsets");
    const int64_t lanes = (n_normal + 3) / 4 + (n_uniform + 1) / 2 + n_subsets;
    const int64_t pb = prologue_span_blocks(n_polyak), zb = prologue_span_blocks(n_zero);
    const float one_m_tau = (float)(1.0 - (double)tau);   // python: (1. - tau) in double, cast by ATen
    // under the measurement repeat knob Polyak (not idempotent) runs in the first repetition only
    for (int rep = 0; rep < g_launch_repeat; ++rep) {
        const int blocks_p = rep == 0 ? (int)pb : 0;
        const PrologueArgs a{seed, step_counter, uniform_out, n_uniform, normal_out, n_normal, subsets_out, n_subsets, E_sample,
                             E, blocks_p, target, source, n_polyak, one_m_tau, tau, (int)zb, zero_out, n_zero};
        hipLaunchKernelGGL(k_noise_fill, dim3((unsigned)(blocks_p + zb + (lanes + 255) / 256)), dim3(256), 0,
                           as_stream(stream), a);
    }
    return finish_launch("asac_step_prologue");
}

// Replays an instantiated hipGraphExec_t on `stream` (the captured train step).
int asac_graph_launch(void* graph_exec, void* stream) {
    if (!graph_exec) return bad_arg("asac_graph_launch");
    const hipError_t e = hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(graph_exec), as_stream(stream));
    if (e != hipSuccess) {
        set_error(e, "asac_graph_launch");
        return (int)e;
    }
    return 0;
}

}  // extern "C"
