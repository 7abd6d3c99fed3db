/* TEST INFRASTRUCTURE ONLY — plain-C restatement of the CTC loss and its gradient.
 *
 * The reference takes this arithmetic from a third-party dependency that is NOT in /root/reference:
 * `warpctc_pytorch` (SeanNaren/warp-ctc binding of baidu-research/warp-ctc); the reference pins no version
 * (no requirements file, lock file or submodule).  Call sites: src/train_cnn_lstm.py:12,358,52,138.
 * This file restates the published algorithm (Graves et al. 2006, "Connectionist Temporal Classification",
 * eqs. 6-16; warp-ctc's cpu_ctc.h computes the same alpha/beta in log space with an internal softmax):
 *   blank = 0, extended label l' of length S = 2L+1,
 *   alpha_t(s) = (alpha_{t-1}(s) + alpha_{t-1}(s-1) + [l'_s != blank && l'_s != l'_{s-2}] alpha_{t-1}(s-2)) * y_t(l'_s)
 *   loss = -log(alpha_T(S-1) + alpha_T(S-2)),  d loss / d a_t(k) = y_t(k) - (1/p) sum_{s: l'_s = k} alpha_t(s) beta_t(s) / y_t(k)
 * in double precision.  tests/test_ctc_c_oracle.py pins it against torch.nn.functional.ctc_loss (the stand-in
 * the north_star names) so the restatement, the PyTorch-CPU path and the HIP kernel are tied together.
 * Only tests/ may call this.  Build: make -C oracle  ->  oracle/_build/libctc_ref.so
 */
#include <math.h>
#include <stdlib.h>

static double lse2(double a, double b) {
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    double m = a > b ? a : b;
    return m + log(exp(a - m) + exp(b - m));
}

/* logits[T][B][V] (pre-softmax), labels flat, label_lens[B], act_lens[B]; nll[B]; grad[T][B][V] (may be NULL). */
int ctc_ref(const float* logits, const int* labels, const int* label_lens, const int* act_lens, int T, int B, int V,
            double* nll, double* grad) {
    int off = 0;
    for (int b = 0; b < B; ++b) {
        const int L = label_lens[b], S = 2 * L + 1, Tb = act_lens[b];
        const int* lab = labels + off;
        off += L;
        double* lp = (double*)malloc(sizeof(double) * (size_t)Tb * V);
        double* al = (double*)malloc(sizeof(double) * (size_t)Tb * S);
        double* be = (double*)malloc(sizeof(double) * (size_t)Tb * S);
        int* ext = (int*)malloc(sizeof(int) * S);
        if (!lp || !al || !be || !ext) return -1;
        for (int s = 0; s < S; ++s) ext[s] = (s & 1) ? lab[s / 2] : 0;
        for (int t = 0; t < Tb; ++t) {                       /* log-softmax over the alphabet */
            const float* x = logits + ((size_t)t * B + b) * V;
            double m = x[0];
            for (int v = 1; v < V; ++v) if (x[v] > m) m = x[v];
            double z = 0;
            for (int v = 0; v < V; ++v) z += exp((double)x[v] - m);
            const double lz = m + log(z);
            for (int v = 0; v < V; ++v) lp[(size_t)t * V + v] = (double)x[v] - lz;
        }
        for (int i = 0; i < Tb * S; ++i) { al[i] = -INFINITY; be[i] = -INFINITY; }
        if (Tb > 0) {
            al[0] = lp[0];
            if (S > 1) al[1] = lp[ext[1]];
            for (int t = 1; t < Tb; ++t)
                for (int s = 0; s < S; ++s) {
                    double a = al[(t - 1) * S + s];
                    if (s >= 1) a = lse2(a, al[(t - 1) * S + s - 1]);
                    if (s >= 2 && ext[s] != 0 && ext[s] != ext[s - 2]) a = lse2(a, al[(t - 1) * S + s - 2]);
                    al[t * S + s] = a == -INFINITY ? a : a + lp[(size_t)t * V + ext[s]];
                }
            be[(Tb - 1) * S + S - 1] = lp[(size_t)(Tb - 1) * V];
            if (S > 1) be[(Tb - 1) * S + S - 2] = lp[(size_t)(Tb - 1) * V + ext[S - 2]];
            for (int t = Tb - 2; t >= 0; --t)
                for (int s = 0; s < S; ++s) {
                    double a = be[(t + 1) * S + s];
                    if (s + 1 < S) a = lse2(a, be[(t + 1) * S + s + 1]);
                    if (s + 2 < S && ext[s] != 0 && ext[s] != ext[s + 2]) a = lse2(a, be[(t + 1) * S + s + 2]);
                    be[t * S + s] = a == -INFINITY ? a : a + lp[(size_t)t * V + ext[s]];
                }
            double ll = al[(Tb - 1) * S + S - 1];
            if (S > 1) ll = lse2(ll, al[(Tb - 1) * S + S - 2]);
            nll[b] = -ll;
        } else {
            nll[b] = S == 1 ? 0.0 : INFINITY;
        }
        if (grad) {
            for (int t = 0; t < T; ++t) {
                double* g = grad + ((size_t)t * B + b) * V;
                for (int v = 0; v < V; ++v) g[v] = 0.0;
                if (t >= Tb) continue;
                for (int v = 0; v < V; ++v) {
                    double acc = -INFINITY;
                    for (int s = 0; s < S; ++s)
                        if (ext[s] == v) acc = lse2(acc, al[t * S + s] + be[t * S + s]);
                    const double y = exp(lp[(size_t)t * V + v]);
                    g[v] = y - (acc == -INFINITY ? 0.0 : exp(acc + nll[b] - lp[(size_t)t * V + v]));
                }
            }
        }
        free(lp); free(al); free(be); free(ext);
    }
    return 0;
}
