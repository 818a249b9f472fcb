// Phase timeline of the fused MLP backward (workgroup (0,0), 100 MHz wall clock): includes the library source with
// ASAC_MLP_STAMPS, drives it through its C ABI at the train step's shapes (N = 256 rows, E = 2 stock Q networks /
// the stock policy) and prints the time between stamps, averaged over launches.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Iadvanced-soft-actor-critic_amd/csrc \
//         tools/mlp_phases.hip -o tools/mlp_phases
#define ASAC_MLP_STAMPS 1
#include "../advanced-soft-actor-critic_amd/csrc/mlp.hip"

#include <cstdio>
#include <vector>

namespace asac {
int g_launch_repeat = 1;
void set_error(hipError_t e, const char* where) { printf("error %s at %s\n", hipGetErrorString(e), where); }
}  // namespace asac

static asac_mlp_desc_t stock(int in0, int in1, int heads0, int heads1, int transform) {
    asac_mlp_desc_t d{};
    d.in0 = in0; d.in1 = in1; d.n_blocks = 3;
    int off = 0, K = in0 + in1;
    for (int l = 0; l < 3; ++l) {
        d.width[l] = 64; d.residual[l] = l > 0;
        d.w_off[l] = off; off += 64 * K;
        d.b_off[l] = off; off += 64;
        K = 64;
    }
    d.head_cols[0] = heads0; d.head_cols[1] = heads1;
    d.head_w_off[0] = off; off += heads0 * 64;
    d.head_b_off[0] = off; off += heads0;
    if (heads1) { d.head_w_off[1] = off; off += heads1 * 64; d.head_b_off[1] = off; off += heads1; }
    d.head_transform = transform;
    return d;
}

int main() {
    const int N = 256, E = 2, A = 2, S = 6;
    const int64_t stride = 9216;
    std::vector<float> h(E * stride);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.05f * (float)((int)(i * 2654435761u >> 20) % 21 - 10) / 10.f;
    float *params, *x0, *x1, *tq, *y, *ws, *grad, *loss, *eps, *ga, *la, *g1, *qt;
    hipMalloc(&params, h.size() * 4); hipMemcpy(params, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&x0, N * 8 * 4); hipMalloc(&x1, N * A * 4); hipMalloc(&tq, E * N * 4); hipMalloc(&y, N * 4);
    hipMalloc(&eps, N * A * 4); hipMalloc(&ga, E * N * A * 4); hipMalloc(&la, 4); hipMalloc(&g1, E * N * A * 4); hipMalloc(&qt, E * N * 4);
    hipMemset(x0, 0, N * 8 * 4); hipMemset(x1, 0, N * A * 4); hipMemset(tq, 0, E * N * 4); hipMemset(y, 0, N * 4);
    hipMemset(eps, 0, N * A * 4); hipMemset(ga, 0, E * N * A * 4); hipMemset(la, 0, 4); hipMemset(qt, 0, E * N * 4);
    const int64_t wsn = asac_mlp_backward_workspace(stride, E, N);
    hipMalloc(&ws, wsn * 4); hipMalloc(&grad, E * stride * 4); hipMalloc(&loss, E * 4);
    const asac_mlp_desc_t dq = stock(S, A, 1, 0, 0), dp = stock(S, 0, A, A, 1);
    for (int mode = 0; mode < 3; ++mode) {
        double acc[24] = {0};
        const int reps = 200;
        for (int r = 0; r < reps; ++r) {
            if (mode == 0)
                asac_mlp_backward_qloss(&dq, params, stride, E, x0, S, 0, x1, A, 0, N, tq, y, nullptr, 0.2f, loss, grad, ws,
                                        ASAC_MLP_REDUCE_DEFER, nullptr);
            else if (mode == 1)
                asac_mlp_backward_policy_q(&dq, params, stride, E, x0, S, 0, x1, A, 0, N, qt, nullptr, E, g1, nullptr);
            else
                asac_mlp_backward_policy_sample(&dp, params, stride, x0, S, N, eps, ga, E, la, grad, ws, ASAC_MLP_REDUCE_DEFER, nullptr);
            hipDeviceSynchronize();
            unsigned long long st[32];
            hipMemcpyFromSymbol(st, HIP_SYMBOL(asac::g_mlp_stamps), sizeof st);
            static const int idx[] = {0, 1, 2, 3, 4, 6, 7, 8, 10};
            for (int i = 1; i < 9; ++i) acc[i] += (double)(st[idx[i]] - st[idx[i - 1]]) / 100.0;
            acc[0] += (double)(st[10] - st[0]) / 100.0;
            acc[10] += (double)(st[16] - st[4]) / 100.0;
            acc[11] += (double)(st[17] - st[16]) / 100.0;
            acc[12] += (double)(st[18] - st[17]) / 100.0;
            acc[13] += (double)(st[6] - st[18]) / 100.0;
            for (int i = 0; i < 4; ++i) acc[14 + i] += (double)(st[21 + i] - st[20 + i]) / 100.0;
        }
        static const char* names[] = {"", "staging", "recompute", "loss / head mode", "head grads + g", "reverse L3", "reverse L2",
                                      "reverse L1", "input grads / end"};
        printf("%s: workgroup (0,0) total %.2f us\n", mode == 0 ? "backward_qloss" : mode == 1 ? "backward_policy_q" : "backward_policy_sample", acc[0] / reps);
        for (int i = 1; i < 9; ++i) printf("   %-18s %6.2f us\n", names[i], acc[i] / reps);
        printf("   recompute layer 2 in detail: gemm %.2f, gelu %.2f, LDS write %.2f, barrier %.2f\n", acc[14] / reps, acc[15] / reps,
               acc[16] / reps, acc[17] / reps);
        if (mode != 1)
            printf("   reverse L3 in detail: delta tile + barriers %.2f, grad_weight %.2f, grad_bias %.2f, dX gemm %.2f\n",
                   acc[10] / reps, acc[11] / reps, acc[12] / reps, acc[13] / reps);
    }
    // ---- the one-launch policy step and the one-launch policy -> sample -> critics forward ----------------------------
    {
        float *act, *qout, *lsout, *aout, *lpout, *prob, *a2, *lp2, *eps5, *x5, *q5;
        const int T = 5, N5 = N * T;
        hipMalloc(&act, N * A * 4); hipMemset(act, 0, N * A * 4); hipMalloc(&qout, E * N * 4);
        hipMalloc(&x5, N5 * 8 * 4); hipMemset(x5, 0, N5 * 8 * 4); hipMalloc(&eps5, N5 * A * 4); hipMemset(eps5, 0, N5 * A * 4);
        hipMalloc(&lsout, N5 * 2 * A * 4); hipMalloc(&aout, N5 * A * 4); hipMalloc(&lpout, N5 * 4); hipMalloc(&prob, N5 * A * 4);
        hipMalloc(&a2, N * A * 4); hipMalloc(&lp2, N * 4); hipMalloc(&q5, E * N5 * 4);
        const int reps = 200;
        double acc[16] = {0};
        for (int r = 0; r < reps; ++r) {
            asac_policy_step_fused(&dq, params, stride, &dp, params, stride, x0, S, N, act, eps, la, nullptr, qout, nullptr, nullptr, nullptr, grad, ws,
                                   ASAC_MLP_REDUCE_DEFER, nullptr);
            hipDeviceSynchronize();
            unsigned long long st[32];
            hipMemcpyFromSymbol(st, HIP_SYMBOL(asac::g_mlp_stamps), sizeof st);
            for (int i = 1; i <= 13; ++i) acc[i] += (double)(st[i] - st[i - 1]) / 100.0;
            acc[0] += (double)(st[13] - st[0]) / 100.0;
        }
        static const char* names[] = {"", "staging", "critics forward", "critic heads", "dq", "critics backward", "policy -> LDS",
                                      "policy forward", "policy head", "sampling backward", "head grads + g", "reverse L3",
                                      "reverse L2", "reverse L1"};
        printf("policy_step_fused: workgroup 0 total %.2f us\n", acc[0] / reps);
        for (int i = 1; i <= 13; ++i) printf("   %-18s %6.2f us\n", names[i], acc[i] / reps);
        asac_pi_q_job_t job{};
        job.pi.desc = &dp; job.pi.params = params; job.pi.member_stride = stride; job.pi.x0 = x5; job.pi.x0_row_stride = S;
        job.pi.N = N5; job.pi.out = lsout; job.pi.E = 1;
        job.q = job.pi; job.q.desc = &dq; job.q.E = E; job.q.out = q5;
        job.sample.eps = eps5; job.sample.a_tanh_out = aout; job.sample.logp_out = lpout; job.sample.rows = N5; job.sample.A = A;
        job.sample.T = T; job.sample.action = aout; job.sample.action_stride_b = T * A; job.sample.action_stride_t = A;
        job.sample.prob_out = prob; job.sample.prob_stride_b = T * A; job.sample.prob_stride_t = A;
        job.eps2 = eps; job.t2 = 0; job.a2_out = a2; job.logp2_out = lp2;
        double acc2[10] = {0};
        for (int r = 0; r < reps; ++r) {
            asac_policy_sample_q_forward(&job, nullptr, 0, nullptr, 0, nullptr);
            hipDeviceSynchronize();
            unsigned long long st[32];
            hipMemcpyFromSymbol(st, HIP_SYMBOL(asac::g_mlp_stamps), sizeof st);
            for (int i = 1; i <= 7; ++i) acc2[i] += (double)(st[i] - st[i - 1]) / 100.0;
            acc2[0] += (double)(st[7] - st[0]) / 100.0;
        }
        static const char* names2[] = {"", "staging", "policy forward", "policy head", "sampling", "actions -> tile",
                                       "critic forward", "critic head"};
        printf("policy_sample_q_forward: workgroup 0 total %.2f us\n", acc2[0] / reps);
        for (int i = 1; i <= 7; ++i) printf("   %-18s %6.2f us\n", names2[i], acc2[i] / reps);
    }
    return 0;
}
