/* Oracle (test infrastructure, never shipped or linked by the product): bit-exact CPU restatement of the fp32
 * similarity score computed by cor_similarity_topk on an fp32 gallery.
 *
 * gfx950's v_mfma_f32_32x32x2_f32 is bitwise a k-ordered fmaf chain, D = fma(a_k1, b_k1, fma(a_k0, b_k0, C))
 * (MI355X guide, "FP32-input MFMA"). The kernel (cor_amd/csrc/retrieval.hip) feeds it 16-byte operand chunks, so for
 * chunk c = 0..C/8-1 and i = 0..3 the chain visits k = 8c+i and then k = 8c+4+i. This file walks the same chain with
 * fmaf(), so scores - and therefore top-k indices - can be compared BITWISE with the GPU.
 * 16-bit galleries: the kernel's final selection re-scores its short list with this very chain over the stored bf16 / fp16
 * values widened to fp32 and the query rounded to the gallery dtype (sim_final in retrieval.hip), so the same functions
 * are the bitwise oracle there too (the caller passes the rounded / widened operands).
 *
 * The reference (wangtong627/COR) has no gallery scoring; the score definition is utils/loss_func.py:84
 * (F.cosine_similarity of unit vectors = dot product). */
#include <math.h>
#include <stddef.h>

void sim_chain_scores(const float* Q, const float* G, int Bq, int Ng, int C, float* out) {
  for (int b = 0; b < Bq; ++b) {
    const float* q = Q + (size_t)b * C;
    for (int g = 0; g < Ng; ++g) {
      const float* r = G + (size_t)g * C;
      float acc = 0.0f;
      for (int c = 0; c < C / 8; ++c)
        for (int i = 0; i < 4; ++i) {
          acc = fmaf(r[8 * c + i], q[8 * c + i], acc);
          acc = fmaf(r[8 * c + 4 + i], q[8 * c + 4 + i], acc);
        }
      out[(size_t)b * Ng + g] = acc;
    }
  }
}

/* Chain scores of selected (query, gallery row) pairs only: out[p] = chain(Q[qi[p]], G[gi[p]]). Lets the tests rank a
 * 1M-row shard exactly without walking 512 x 1M chains: rows whose plain fp32 score is far below the k-th best cannot enter
 * the chain top-k (the two summation orders differ by < 3.1e-5 for unit vectors), so only the near-top rows are chained. */
void sim_chain_pairs(const float* Q, const float* G, const long long* qi, const long long* gi, long long n, int C, float* out) {
  for (long long p = 0; p < n; ++p) {
    const float* q = Q + (size_t)qi[p] * C;
    const float* r = G + (size_t)gi[p] * C;
    float acc = 0.0f;
    for (int c = 0; c < C / 8; ++c)
      for (int i = 0; i < 4; ++i) {
        acc = fmaf(r[8 * c + i], q[8 * c + i], acc);
        acc = fmaf(r[8 * c + 4 + i], q[8 * c + 4 + i], acc);
      }
    out[p] = acc;
  }
}
