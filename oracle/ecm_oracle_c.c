/*
 * CPU ORACLE, plain C -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * A framework-independent restatement of the reference's hot-path arithmetic as naive loops, used by
 * tests/test_oracle_c.py to cross-check oracle/ecm_oracle.py (which shares ATen with the reference) and the golden
 * vectors.  fp32 storage, double accumulation.  Reference lines: cmf/models/cmfsm.py.
 */
#include <math.h>
#include <stddef.h>

/* a1: cost volume, cmfsm.py:667-682.  L,R [B,C,h,w] -> cost [B,2C,D,h,w] */
void oc_cost_volume(const float* L, const float* R, float* cost, int B, int C, int h, int w, int D) {
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < 2 * C; ++c)
            for (int d = 0; d < D; ++d)
                for (int y = 0; y < h; ++y)
                    for (int x = 0; x < w; ++x) {
                        float v = 0.f;
                        if (x >= d) {
                            if (c < C) v = L[((size_t)(b * C + c) * h + y) * w + x];
                            else v = R[((size_t)(b * C + c - C) * h + y) * w + x - d];
                        }
                        cost[((((size_t)b * 2 * C + c) * D + d) * h + y) * w + x] = v;
                    }
}

/* a8: softmax over D + disparity regression, cmfsm.py:703-706, 111-123.  cost [B,D,hw] -> disp [B,hw] */
void oc_soft_argmin(const float* cost, float* disp, int B, int D, int hw) {
    for (int b = 0; b < B; ++b)
        for (int p = 0; p < hw; ++p) {
            double m = -INFINITY, s = 0.0, t = 0.0;
            for (int d = 0; d < D; ++d) { double v = cost[((size_t)b * D + d) * hw + p]; if (v > m) m = v; }
            for (int d = 0; d < D; ++d) { double e = exp((double)cost[((size_t)b * D + d) * hw + p] - m); s += e; t += e * d; }
            disp[(size_t)b * hw + p] = (float)(t / s);
        }
}

static const int DY9[9] = {0, 0, 0, -1, 1, -1, -1, 1, 1};
static const int DX9[9] = {0, -1, 1, 0, 0, -1, 1, -1, 1};
static const int TAB9[9] = {0, 1, 2, 3, 4, 1, 2, 3, 4};      /* cmfsm.py:459-462: tables 5..8 alias 1..4 */

static double centre_pat(int r, int s) { return r < s / 2 ? r - s / 2 : r - s / 2 + 1; }
static double offx(int t, int r, int s) { return t == 1 ? s - r : t == 2 ? r + 1 : centre_pat(r, s); }
static double offy(int t, int r, int s) { return t == 3 ? s - r : t == 4 ? r + 1 : centre_pat(r, s); }
static double lrelu(double x) { return x > 0 ? x : 0.01 * x; }

/* a3: eight_related_context_mapping, cmfsm.py:431-593.  lr [B,32,h,w], hr [B,32,H,W] -> w9 [B,9,H,W] */
void oc_ecm_weights_eight(const float* lr, const float* hr, const float* W0, const float* W1, const float* W2,
                          const float* W3, float* w9, int B, int h, int w, int s) {
    const int H = h * s, W = w * s;
    for (int b = 0; b < B; ++b)
        for (int Y = 0; Y < H; ++Y)
            for (int X = 0; X < W; ++X) {
                double logit[9];
                for (int n = 0; n < 9; ++n) {
                    const int cy = Y / s + DY9[n], cx = X / s + DX9[n];
                    if (cy < 0 || cy >= h || cx < 0 || cx >= w) { logit[n] = -100.0; continue; }
                    double in[66], h0[32], h1[16], h2[8];
                    for (int c = 0; c < 32; ++c) {
                        in[c] = lr[((size_t)(b * 32 + c) * h + cy) * w + cx];
                        in[32 + c] = hr[((size_t)(b * 32 + c) * H + Y) * W + X];
                    }
                    in[64] = offx(TAB9[n], X % s, s);
                    in[65] = offy(TAB9[n], Y % s, s);
                    for (int j = 0; j < 32; ++j) { double a = 0; for (int c = 0; c < 66; ++c) a += (double)W0[j * 66 + c] * in[c]; h0[j] = lrelu(a); }
                    for (int j = 0; j < 16; ++j) { double a = 0; for (int c = 0; c < 32; ++c) a += (double)W1[j * 32 + c] * h0[c]; h1[j] = lrelu(a); }
                    for (int j = 0; j < 8; ++j) { double a = 0; for (int c = 0; c < 16; ++c) a += (double)W2[j * 16 + c] * h1[c]; h2[j] = lrelu(a); }
                    double o = 0; for (int c = 0; c < 8; ++c) o += (double)W3[c] * h2[c];
                    logit[n] = o;
                }
                double m = logit[0], sum = 0;
                for (int n = 1; n < 9; ++n) if (logit[n] > m) m = logit[n];
                for (int n = 0; n < 9; ++n) sum += exp(logit[n] - m);
                for (int n = 0; n < 9; ++n) w9[(((size_t)b * 9 + n) * H + Y) * W + X] = (float)(exp(logit[n] - m) / sum);
            }
}

/* a9: NN-upsample x scale + 9-neighbour weighted sum, cmfsm.py:709-723.  d [B,h,w], w9 [B,9,H,W] -> out [B,H,W] */
void oc_aggregate9(const float* d, const float* w9, float* out, int B, int h, int w, int s) {
    const int H = h * s, W = w * s;
    for (int b = 0; b < B; ++b)
        for (int Y = 0; Y < H; ++Y)
            for (int X = 0; X < W; ++X) {
                double acc = 0;
                for (int n = 0; n < 9; ++n) {
                    const int cy = Y / s + DY9[n], cx = X / s + DX9[n];
                    if (cy < 0 || cy >= h || cx < 0 || cx >= w) continue;
                    acc += (double)w9[(((size_t)b * 9 + n) * H + Y) * W + X] * (s * (double)d[((size_t)b * h + cy) * w + cx]);
                }
                out[((size_t)b * H + Y) * W + X] = (float)acc;
            }
}

/* nn.Conv3d(k=3, pad=1, stride, bias=False), cmfsm.py:52-57.  x [B,Ci,D,H,W], wt [Co,Ci,27] -> y [B,Co,Do,Ho,Wo] */
void oc_conv3d_k3(const float* x, const float* wt, float* y, int B, int Ci, int Co, int D, int H, int W, int st) {
    const int Do = (D - 1) / st + 1, Ho = (H - 1) / st + 1, Wo = (W - 1) / st + 1;
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Co; ++co)
            for (int od = 0; od < Do; ++od)
                for (int oh = 0; oh < Ho; ++oh)
                    for (int ow = 0; ow < Wo; ++ow) {
                        double acc = 0;
                        for (int ci = 0; ci < Ci; ++ci)
                            for (int kd = 0; kd < 3; ++kd)
                                for (int kh = 0; kh < 3; ++kh)
                                    for (int kw = 0; kw < 3; ++kw) {
                                        const int z = od * st + kd - 1, yy = oh * st + kh - 1, xx = ow * st + kw - 1;
                                        if (z < 0 || z >= D || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                                        acc += (double)wt[((size_t)co * Ci + ci) * 27 + (kd * 3 + kh) * 3 + kw] *
                                               x[((((size_t)b * Ci + ci) * D + z) * H + yy) * W + xx];
                                    }
                        y[((((size_t)b * Co + co) * Do + od) * Ho + oh) * Wo + ow] = (float)acc;
                    }
}

/* nn.ConvTranspose3d(k=3, stride=2, pad=1, output_padding=1, bias=False), cmfsm.py:262-268.
 * x [B,Ci,D,H,W], wt [Ci,Co,27] -> y [B,Co,2D,2H,2W];  y[o] += x[i] * w[k] for o = 2i - 1 + k */
void oc_deconv3d_k3s2(const float* x, const float* wt, float* y, int B, int Ci, int Co, int D, int H, int W) {
    const int Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
    for (size_t i = 0; i < (size_t)B * Co * Do * Ho * Wo; ++i) y[i] = 0.f;
    for (int b = 0; b < B; ++b)
        for (int ci = 0; ci < Ci; ++ci)
            for (int z = 0; z < D; ++z)
                for (int yy = 0; yy < H; ++yy)
                    for (int xx = 0; xx < W; ++xx) {
                        const float xv = x[((((size_t)b * Ci + ci) * D + z) * H + yy) * W + xx];
                        for (int co = 0; co < Co; ++co)
                            for (int kd = 0; kd < 3; ++kd)
                                for (int kh = 0; kh < 3; ++kh)
                                    for (int kw = 0; kw < 3; ++kw) {
                                        const int od = 2 * z - 1 + kd, oh = 2 * yy - 1 + kh, ow = 2 * xx - 1 + kw;
                                        if (od < 0 || od >= Do || oh < 0 || oh >= Ho || ow < 0 || ow >= Wo) continue;
                                        y[((((size_t)b * Co + co) * Do + od) * Ho + oh) * Wo + ow] +=
                                            xv * wt[((size_t)ci * Co + co) * 27 + (kd * 3 + kh) * 3 + kw];
                                    }
                    }
}

/* nn.GroupNorm(32, C, eps=1e-5) [+ skip] [+ ReLU], cmfsm.py:58, 287-299.  x,y [B,C,S]; skip may be NULL */
void oc_group_norm(const float* x, const float* gamma, const float* beta, const float* skip, float* y, int B, int C,
                   long S, int relu) {
    const int G = 32, cpg = C / G;
    for (int b = 0; b < B; ++b)
        for (int g = 0; g < G; ++g) {
            const size_t base = ((size_t)b * C + (size_t)g * cpg) * S, n = (size_t)cpg * S;
            double s = 0, q = 0;
            for (size_t i = 0; i < n; ++i) { s += x[base + i]; }
            const double mean = s / n;
            for (size_t i = 0; i < n; ++i) { const double d = x[base + i] - mean; q += d * d; }
            const double rstd = 1.0 / sqrt(q / n + 1e-5);
            for (int j = 0; j < cpg; ++j)
                for (long i = 0; i < S; ++i) {
                    const size_t idx = base + (size_t)j * S + i;
                    double v = (x[idx] - mean) * rstd * gamma[g * cpg + j] + beta[g * cpg + j];
                    if (skip) v += skip[idx];
                    if (relu && v < 0) v = 0;
                    y[idx] = (float)v;
                }
        }
}
