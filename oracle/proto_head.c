/* ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Plain-C restatement of the reference's prototype head with double accumulation, used by the tests as an
 * independent high-precision check of both the PyTorch-CPU restatement (oracle/blocks.py) and the HIP kernels.
 *   proto_configure_ref : model/classifier_heads.py:94-119 (_build_class_reps: per-class mean of the rows whose
 *                         label equals the class) + :232-263 (W = 2 mu, b = -mu.mu; cosine: W only)
 *   proto_predict_ref   : model/classifier_heads.py:202-219 (euclidean: s*(q.W + b); cosine: s*cos(q, W), eps 1e-8)
 * Built by `make -C oracle` (gcc) into oracle/_build/libproto_head_ref.so.
 */
#include <math.h>
#include <stdint.h>

/* class_ids: ascending unique labels [C]. Returns 0, or -1 if a class has no rows. */
int proto_configure_ref(const float* feats, const int64_t* labels, const int64_t* class_ids, int N, int D, int C,
                        int cosine, float* W, float* b) {
    for (int c = 0; c < C; ++c) {
        int cnt = 0;
        for (int i = 0; i < N; ++i) cnt += labels[i] == class_ids[c];
        if (cnt == 0) return -1;
        double sq = 0.0;
        for (int d = 0; d < D; ++d) {
            double s = 0.0;
            for (int i = 0; i < N; ++i)
                if (labels[i] == class_ids[c]) s += feats[(int64_t)i * D + d];
            const double mu = s / cnt;
            W[(int64_t)c * D + d] = (float)(2.0 * mu);
            sq += mu * mu;
        }
        if (!cosine) b[c] = (float)(-sq);
    }
    return 0;
}

void proto_predict_ref(const float* Q, const float* W, const float* b, int M, int D, int C, float logit_scale,
                       int cosine, float* logits) {
    for (int m = 0; m < M; ++m) {
        double qn = 0.0;
        for (int d = 0; d < D; ++d) qn += (double)Q[(int64_t)m * D + d] * Q[(int64_t)m * D + d];
        qn = sqrt(qn);
        for (int c = 0; c < C; ++c) {
            double dot = 0.0, wn = 0.0;
            for (int d = 0; d < D; ++d) {
                const double w = W[(int64_t)c * D + d];
                dot += (double)Q[(int64_t)m * D + d] * w;
                wn += w * w;
            }
            if (cosine) {
                wn = sqrt(wn);
                const double den = (qn > 1e-8 ? qn : 1e-8) * (wn > 1e-8 ? wn : 1e-8);
                logits[(int64_t)m * C + c] = (float)(logit_scale * dot / den);
            } else {
                logits[(int64_t)m * C + c] = (float)(logit_scale * (dot + b[c]));
            }
        }
    }
}
