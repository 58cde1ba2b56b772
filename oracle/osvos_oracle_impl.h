/* ORACLE (test infrastructure only) -- plain-C restatement of the OSVOS hot path, NCHW.
 * Included twice by osvos_oracle.c with REAL = float / double and SUF(x) = x##_f32 / x##_f64.
 * Storage is REAL (each layer's output is rounded to REAL like the reference's tensors); every
 * dot product / reduction accumulates in double.  Not a product path: nothing under
 * osvos-pytorch_amd/ may link or call this.
 *
 * Semantics restated (the arithmetic itself lives in PyTorch/ATen, unpinned -- see
 * SURVEY.md 8c; definitions below follow SURVEY.md Appendix D and the reference call sites):
 *   conv3x3 / conv1x1 ....... nn.Conv2d, cross-correlation, zero pad  (vgg_osvos.py:41,44,54,142)
 *   maxpool 2x2/2 ceil ...... nn.MaxPool2d(ceil_mode=True), first-max-wins backward (vgg_osvos.py:140)
 *   transposed conv ......... nn.ConvTranspose2d(k=2s, stride=s, no pad, no bias) (vgg_osvos.py:45-46)
 *   center crop ............. negative F.pad (osvos_layers.py:51-56)
 *   class-balanced BCE ...... osvos_layers.py:19-48
 *   forward wiring .......... vgg_osvos.py:59-74
 */

static void SUF(conv_fwd)(const REAL* x, const REAL* w, const REAL* b, REAL* y,
                          int N, int Cin, int H, int W, int Cout, int K, int relu) {
  const int P = K / 2;
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < Cout; ++co) {
      double* acc = (double*)malloc(sizeof(double) * W);
      for (int yy = 0; yy < H; ++yy) {
        for (int xx = 0; xx < W; ++xx) acc[xx] = b ? (double)b[co] : 0.0;
        for (int ci = 0; ci < Cin; ++ci)
          for (int r = 0; r < K; ++r) {
            int iy = yy + r - P;
            if (iy < 0 || iy >= H) continue;
            const REAL* xr = x + (((size_t)n * Cin + ci) * H + iy) * W;
            for (int s = 0; s < K; ++s) {
              double wv = (double)w[(((size_t)co * Cin + ci) * K + r) * K + s];
              int x0 = P - s > 0 ? P - s : 0;
              int x1 = W + P - s < W ? W + P - s : W;
              for (int xx = x0; xx < x1; ++xx) acc[xx] += wv * (double)xr[xx + s - P];
            }
          }
        REAL* yr = y + (((size_t)n * Cout + co) * H + yy) * W;
        for (int xx = 0; xx < W; ++xx) {
          REAL v = (REAL)acc[xx];
          yr[xx] = (relu && !(v > 0)) ? (REAL)0 : v;
        }
      }
      free(acc);
    }
}

/* dx (may be NULL), dw, db (may be NULL) are OVERWRITTEN.  dy must already carry the ReLU mask. */
static void SUF(conv_bwd)(const REAL* x, const REAL* w, const REAL* dy, REAL* dx, REAL* dw, REAL* db,
                          int N, int Cin, int H, int W, int Cout, int K) {
  const int P = K / 2;
  if (dx) {
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int n = 0; n < N; ++n)
      for (int ci = 0; ci < Cin; ++ci) {
        double* acc = (double*)malloc(sizeof(double) * W);
        for (int iy = 0; iy < H; ++iy) {
          for (int xx = 0; xx < W; ++xx) acc[xx] = 0.0;
          for (int co = 0; co < Cout; ++co)
            for (int r = 0; r < K; ++r) {
              int oy = iy - r + P;                 /* output row that read input row iy with tap r */
              if (oy < 0 || oy >= H) continue;
              const REAL* dr = dy + (((size_t)n * Cout + co) * H + oy) * W;
              for (int s = 0; s < K; ++s) {
                double wv = (double)w[(((size_t)co * Cin + ci) * K + r) * K + s];
                /* input col ix was read by output col ox = ix - s + P */
                int x0 = s - P > 0 ? s - P : 0;
                int x1 = W + s - P < W ? W + s - P : W;
                for (int ix = x0; ix < x1; ++ix) acc[ix] += wv * (double)dr[ix - s + P];
              }
            }
          REAL* o = dx + (((size_t)n * Cin + ci) * H + iy) * W;
          for (int xx = 0; xx < W; ++xx) o[xx] = (REAL)acc[xx];
        }
        free(acc);
      }
  }
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int co = 0; co < Cout; ++co)
    for (int ci = 0; ci < Cin; ++ci) {
      double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < H; ++oy) {
          const REAL* dr = dy + (((size_t)n * Cout + co) * H + oy) * W;
          for (int r = 0; r < K; ++r) {
            int iy = oy + r - P;
            if (iy < 0 || iy >= H) continue;
            const REAL* xr = x + (((size_t)n * Cin + ci) * H + iy) * W;
            for (int s = 0; s < K; ++s) {
              int x0 = P - s > 0 ? P - s : 0;
              int x1 = W + P - s < W ? W + P - s : W;
              double t = 0.0;
              for (int ox = x0; ox < x1; ++ox) t += (double)dr[ox] * (double)xr[ox + s - P];
              a[r * K + s] += t;
            }
          }
        }
      for (int t = 0; t < K * K; ++t) dw[((size_t)co * Cin + ci) * K * K + t] = (REAL)a[t];
    }
  if (db) {
#pragma omp parallel for
    for (int co = 0; co < Cout; ++co) {
      double t = 0.0;
      for (int n = 0; n < N; ++n) {
        const REAL* dr = dy + ((size_t)n * Cout + co) * H * W;
        for (int i = 0; i < H * W; ++i) t += (double)dr[i];
      }
      db[co] = (REAL)t;
    }
  }
}

/* y: [N,C,ceil(H/2),ceil(W/2)]; arg: flat input index (iy*W+ix) of the first max in scan order */
static void SUF(pool_fwd)(const REAL* x, REAL* y, int* arg, int N, int C, int H, int W) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
#pragma omp parallel for
  for (int nc = 0; nc < N * C; ++nc) {
    const REAL* xp = x + (size_t)nc * H * W;
    for (int oy = 0; oy < Ho; ++oy)
      for (int ox = 0; ox < Wo; ++ox) {
        int best = (2 * oy) * W + 2 * ox;
        REAL bv = xp[best];
        for (int dyy = 0; dyy < 2; ++dyy)
          for (int dxx = 0; dxx < 2; ++dxx) {
            int iy = 2 * oy + dyy, ix = 2 * ox + dxx;
            if (iy >= H || ix >= W) continue;          /* clipped window, never padded */
            REAL v = xp[iy * W + ix];
            if (v > bv || v != v) { bv = v; best = iy * W + ix; }
          }
        y[((size_t)nc * Ho + oy) * Wo + ox] = bv;
        arg[((size_t)nc * Ho + oy) * Wo + ox] = best;
      }
  }
}

/* dx is ACCUMULATED INTO (caller zeroes or pre-loads the other branch's gradient) */
static void SUF(pool_bwd_acc)(const REAL* dy, const int* arg, REAL* dx, int N, int C, int H, int W) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
#pragma omp parallel for
  for (int nc = 0; nc < N * C; ++nc)
    for (int i = 0; i < Ho * Wo; ++i)
      dx[(size_t)nc * H * W + arg[(size_t)nc * Ho * Wo + i]] += dy[(size_t)nc * Ho * Wo + i];
}

/* ConvTranspose2d(Cin,Cout,k,stride=s), weight [Cin][Cout][k][k], output (h-1)*s+k */
static void SUF(deconv_fwd)(const REAL* x, const REAL* w, REAL* y, int N, int Cin, int Cout, int h, int wd, int k, int s) {
  const int Ho = (h - 1) * s + k, Wo = (wd - 1) * s + k;
#pragma omp parallel for collapse(2)
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < Cout; ++co)
      for (int Y = 0; Y < Ho; ++Y)
        for (int X = 0; X < Wo; ++X) {
          double t = 0.0;
          for (int ci = 0; ci < Cin; ++ci)
            for (int yy = 0; yy < h; ++yy) {
              int ky = Y - yy * s;
              if (ky < 0 || ky >= k) continue;
              for (int xx = 0; xx < wd; ++xx) {
                int kx = X - xx * s;
                if (kx < 0 || kx >= k) continue;
                t += (double)x[(((size_t)n * Cin + ci) * h + yy) * wd + xx] *
                     (double)w[(((size_t)ci * Cout + co) * k + ky) * k + kx];
              }
            }
          y[(((size_t)n * Cout + co) * Ho + Y) * Wo + X] = (REAL)t;
        }
}

/* gradient w.r.t. the deconv input (the weights are frozen, lr 0: train_online.py:84-85) */
static void SUF(deconv_bwd_in)(const REAL* dy, const REAL* w, REAL* dx, int N, int Cin, int Cout, int h, int wd, int k, int s) {
  const int Ho = (h - 1) * s + k, Wo = (wd - 1) * s + k;
#pragma omp parallel for collapse(2)
  for (int n = 0; n < N; ++n)
    for (int ci = 0; ci < Cin; ++ci)
      for (int yy = 0; yy < h; ++yy)
        for (int xx = 0; xx < wd; ++xx) {
          double t = 0.0;
          for (int co = 0; co < Cout; ++co)
            for (int ky = 0; ky < k; ++ky)
              for (int kx = 0; kx < k; ++kx)
                t += (double)dy[(((size_t)n * Cout + co) * Ho + yy * s + ky) * Wo + xx * s + kx] *
                     (double)w[(((size_t)ci * Cout + co) * k + ky) * k + kx];
          dx[(((size_t)n * Cin + ci) * h + yy) * wd + xx] = (REAL)t;
        }
}

/* crop offsets: floor(excess/2) removed top/left (osvos_layers.py:52-56) */
static void SUF(crop_fwd)(const REAL* x, REAL* y, int NC, int Hi, int Wi, int H, int W) {
  const int t = (Hi - H) / 2, l = (Wi - W) / 2;
  for (int c = 0; c < NC; ++c)
    for (int yy = 0; yy < H; ++yy)
      memcpy(y + ((size_t)c * H + yy) * W, x + ((size_t)c * Hi + yy + t) * Wi + l, sizeof(REAL) * W);
}
static void SUF(crop_bwd)(const REAL* dy, REAL* dx, int NC, int Hi, int Wi, int H, int W) {
  const int t = (Hi - H) / 2, l = (Wi - W) / 2;
  memset(dx, 0, sizeof(REAL) * (size_t)NC * Hi * Wi);
  for (int c = 0; c < NC; ++c)
    for (int yy = 0; yy < H; ++yy)
      memcpy(dx + ((size_t)c * Hi + yy + t) * Wi + l, dy + ((size_t)c * H + yy) * W, sizeof(REAL) * W);
}

/* class-balanced BCE with logits (osvos_layers.py:19-48).  mode: 0 size_average, 1 batch_average,
 * 2 none.  grad (may be NULL) receives dLoss/dOutput for an upstream gradient of 1. */
double SUF(osvos_oracle_cbce)(const REAL* out, const REAL* label, REAL* grad, int N, long per_image, int mode) {
  const long total = (long)N * per_image;
  double npos = 0.0;
  for (long i = 0; i < total; ++i) npos += (label[i] >= (REAL)0.5) ? 1.0 : 0.0;
  const double nneg = (double)total - npos, ntot = (double)total;
  /* the reference forms both class weights as float32 quotients (masks are cast with .float(),
   * osvos_layers.py:28-32,41), whatever the dtype of the logits */
  const double wpos = (double)((float)nneg / (float)ntot), wneg = (double)((float)npos / (float)ntot);
  double lpos = 0.0, lneg = 0.0;
  for (long i = 0; i < total; ++i) {
    double o = (double)out[i];
    double y = (label[i] >= (REAL)0.5) ? 1.0 : 0.0;
    double g = (o >= 0) ? 1.0 : 0.0;
    double val = o * (y - g) - log(1.0 + exp(o - 2.0 * o * g));
    lpos += -y * val;
    lneg += -(1.0 - y) * val;
  }
  double div = mode == 0 ? (double)total : (mode == 1 ? (double)N : 1.0);
  if (grad) {
    for (long i = 0; i < total; ++i) {
      double o = (double)out[i];
      double y = (label[i] >= (REAL)0.5) ? 1.0 : 0.0;
      double sg = 1.0 / (1.0 + exp(-o));
      double wgt = y > 0.5 ? wpos : wneg;
      grad[i] = (REAL)(wgt * (sg - y) / div);
    }
  }
  return (wpos * lpos + wneg * lneg) / div;
}

/* ---- whole network ------------------------------------------------------------------------
 * params: 52 pointers in the reference's state_dict order:
 *   [0..3] upscale.i.weight  [4..7] upscale_.i.weight  [8..33] trunk (w,b)x13
 *   [34..41] side_prep (w,b)x4  [42..49] score_dsn (w,b)x4  [50] fuse.weight [51] fuse.bias
 * outs: 5 maps [N,1,H,W].  If label != NULL also computes the per-head losses (size_average=False,
 * i.e. divided by N) and, if grads != NULL, the gradient of
 *     loss_scale * ( side_w * sum(losses[0..3]) + losses[4] )
 * w.r.t. every parameter (grads[52], each sized like its parameter; deconv weight grads are
 * written as zeros: they are frozen) and w.r.t. the input (dx, may be NULL). */
static const int SUF(kStageN)[5] = {2, 2, 3, 3, 3};
static const int SUF(kStageC)[5] = {64, 128, 256, 512, 512};

int SUF(osvos_oracle_net)(const REAL* const* params, const REAL* x, const REAL* label, int N, int H, int W,
                          double side_w, double loss_scale, REAL* const* outs, double* losses,
                          REAL* const* grads, REAL* dx) {
  int hs[5], ws[5];
  hs[0] = H; ws[0] = W;
  for (int i = 1; i < 5; ++i) { hs[i] = (hs[i - 1] + 1) / 2; ws[i] = (ws[i - 1] + 1) / 2; }
  /* activations: act[si][j] output of conv j of stage si (post-ReLU); pooled[si] input of stage si */
  REAL* act[5][3]; REAL* pooled[5]; int* arg[5];
  REAL* prep[4]; REAL* score[4]; REAL* up16[4]; REAL* up1[4];
  memset(act, 0, sizeof(act)); memset(pooled, 0, sizeof(pooled)); memset(arg, 0, sizeof(arg));
  const REAL* cur = x; int ccur = 3; int pi = 8;
  for (int si = 0; si < 5; ++si) {
    const size_t hw = (size_t)hs[si] * ws[si];
    if (si > 0) {
      pooled[si] = (REAL*)malloc(sizeof(REAL) * N * ccur * hw);
      arg[si] = (int*)malloc(sizeof(int) * N * ccur * hw);
      SUF(pool_fwd)(cur, pooled[si], arg[si], N, ccur, hs[si - 1], ws[si - 1]);
      cur = pooled[si];
    }
    for (int j = 0; j < SUF(kStageN)[si]; ++j) {
      act[si][j] = (REAL*)malloc(sizeof(REAL) * N * SUF(kStageC)[si] * hw);
      SUF(conv_fwd)(cur, params[pi], params[pi + 1], act[si][j], N, ccur, hs[si], ws[si], SUF(kStageC)[si], 3, 1);
      pi += 2; cur = act[si][j]; ccur = SUF(kStageC)[si];
    }
  }
  REAL* cat = (REAL*)malloc(sizeof(REAL) * (size_t)N * 64 * H * W);
  for (int i = 0; i < 4; ++i) {
    const int si = i + 1, s = 1 << si, k = 2 * s, h = hs[si], w = ws[si];
    const int Ho = (h - 1) * s + k, Wo = (w - 1) * s + k;
    const REAL* xin = act[si][SUF(kStageN)[si] - 1];
    prep[i] = (REAL*)malloc(sizeof(REAL) * (size_t)N * 16 * h * w);
    SUF(conv_fwd)(xin, params[34 + 2 * i], params[35 + 2 * i], prep[i], N, SUF(kStageC)[si], h, w, 16, 3, 0);
    up16[i] = (REAL*)malloc(sizeof(REAL) * (size_t)N * 16 * Ho * Wo);
    SUF(deconv_fwd)(prep[i], params[i], up16[i], N, 16, 16, h, w, k, s);
    for (int n = 0; n < N; ++n)   /* cat(dim=1): channel = 16*i + c  (vgg_osvos.py:71) */
      SUF(crop_fwd)(up16[i] + (size_t)n * 16 * Ho * Wo, cat + ((size_t)n * 64 + 16 * i) * H * W, 16, Ho, Wo, H, W);
    score[i] = (REAL*)malloc(sizeof(REAL) * (size_t)N * h * w);
    SUF(conv_fwd)(prep[i], params[42 + 2 * i], params[43 + 2 * i], score[i], N, 16, h, w, 1, 1, 0);
    up1[i] = (REAL*)malloc(sizeof(REAL) * (size_t)N * Ho * Wo);
    SUF(deconv_fwd)(score[i], params[4 + i], up1[i], N, 1, 1, h, w, k, s);
    SUF(crop_fwd)(up1[i], outs[i], N, Ho, Wo, H, W);
  }
  SUF(conv_fwd)(cat, params[50], params[51], outs[4], N, 64, H, W, 1, 1, 0);

  if (label) {
    REAL* dout[5];
    for (int i = 0; i < 5; ++i) {
      dout[i] = grads ? (REAL*)malloc(sizeof(REAL) * (size_t)N * H * W) : NULL;
      losses[i] = SUF(osvos_oracle_cbce)(outs[i], label, dout[i], N, (long)H * W, 1);
      if (dout[i]) {
        double sc = loss_scale * (i < 4 ? side_w : 1.0);
        for (size_t t = 0; t < (size_t)N * H * W; ++t) dout[i][t] = (REAL)(sc * (double)dout[i][t]);
      }
    }
    if (grads) {
      /* fuse backward */
      REAL* dcat = (REAL*)malloc(sizeof(REAL) * (size_t)N * 64 * H * W);
      SUF(conv_bwd)(cat, params[50], dout[4], dcat, grads[50], grads[51], N, 64, H, W, 1, 1);
      REAL* dstage_out[5] = {0, 0, 0, 0, 0};  /* gradient reaching act[si][last] from the side branch */
      for (int i = 0; i < 4; ++i) {
        const int si = i + 1, s = 1 << si, k = 2 * s, h = hs[si], w = ws[si];
        const int Ho = (h - 1) * s + k, Wo = (w - 1) * s + k;
        REAL* dup16 = (REAL*)malloc(sizeof(REAL) * (size_t)N * 16 * Ho * Wo);
        for (int n = 0; n < N; ++n)
          SUF(crop_bwd)(dcat + ((size_t)n * 64 + 16 * i) * H * W, dup16 + (size_t)n * 16 * Ho * Wo, 16, Ho, Wo, H, W);
        REAL* dprep = (REAL*)malloc(sizeof(REAL) * (size_t)N * 16 * h * w);
        SUF(deconv_bwd_in)(dup16, params[i], dprep, N, 16, 16, h, w, k, s);
        REAL* dup1 = (REAL*)malloc(sizeof(REAL) * (size_t)N * Ho * Wo);
        SUF(crop_bwd)(dout[i], dup1, N, Ho, Wo, H, W);
        REAL* dscore = (REAL*)malloc(sizeof(REAL) * (size_t)N * h * w);
        SUF(deconv_bwd_in)(dup1, params[4 + i], dscore, N, 1, 1, h, w, k, s);
        REAL* dprep2 = (REAL*)malloc(sizeof(REAL) * (size_t)N * 16 * h * w);
        SUF(conv_bwd)(prep[i], params[42 + 2 * i], dscore, dprep2, grads[42 + 2 * i], grads[43 + 2 * i], N, 16, h, w, 1, 1);
        for (size_t t = 0; t < (size_t)N * 16 * h * w; ++t) dprep[t] = (REAL)((double)dprep[t] + (double)dprep2[t]);
        dstage_out[si] = (REAL*)malloc(sizeof(REAL) * (size_t)N * SUF(kStageC)[si] * h * w);
        SUF(conv_bwd)(act[si][SUF(kStageN)[si] - 1], params[34 + 2 * i], dprep, dstage_out[si],
                      grads[34 + 2 * i], grads[35 + 2 * i], N, SUF(kStageC)[si], h, w, 16, 3);
        memset(grads[i], 0, sizeof(REAL) * (size_t)16 * 16 * k * k);
        memset(grads[4 + i], 0, sizeof(REAL) * (size_t)k * k);
        free(dup16); free(dprep); free(dup1); free(dscore); free(dprep2);
      }
      /* trunk backward, deepest stage first */
      REAL* g = dstage_out[4];   /* gradient w.r.t. act[4][last] (only the side branch feeds it) */
      pi = 8 + 2 * 13;
      for (int si = 4; si >= 0; --si) {
        const int C = SUF(kStageC)[si], h = hs[si], w = ws[si];
        for (int j = SUF(kStageN)[si] - 1; j >= 0; --j) {
          pi -= 2;
          const size_t cnt = (size_t)N * C * h * w;
          for (size_t t = 0; t < cnt; ++t) if (!(act[si][j][t] > 0)) g[t] = 0;   /* ReLU backward */
          const REAL* xin = j > 0 ? act[si][j - 1] : (si > 0 ? pooled[si] : x);
          const int cin = j > 0 ? C : (si > 0 ? SUF(kStageC)[si - 1] : 3);
          int need_dx = !(si == 0 && j == 0) || dx != NULL;
          REAL* gin = need_dx ? (REAL*)malloc(sizeof(REAL) * (size_t)N * cin * h * w) : NULL;
          SUF(conv_bwd)(xin, params[pi], g, gin, grads[pi], grads[pi + 1], N, cin, h, w, C, 3);
          free(g); g = gin;
        }
        if (si > 0) {   /* through the pool into act[si-1][last], plus that stage's side branch */
          const int Cp = SUF(kStageC)[si - 1];
          const size_t cnt = (size_t)N * Cp * hs[si - 1] * ws[si - 1];
          REAL* gprev = dstage_out[si - 1];
          if (!gprev) gprev = (REAL*)calloc(cnt, sizeof(REAL));
          SUF(pool_bwd_acc)(g, arg[si], gprev, N, Cp, hs[si - 1], ws[si - 1]);
          free(g); g = gprev;
        }
      }
      if (dx && g) memcpy(dx, g, sizeof(REAL) * (size_t)N * 3 * H * W);
      free(g); free(dcat);
    }
    for (int i = 0; i < 5; ++i) free(dout[i]);
  }
  for (int si = 0; si < 5; ++si) {
    for (int j = 0; j < 3; ++j) free(act[si][j]);
    free(pooled[si]); free(arg[si]);
  }
  for (int i = 0; i < 4; ++i) { free(prep[i]); free(score[i]); free(up16[i]); free(up1[i]); }
  free(cat);
  return 0;
}

/* thin exported wrappers for per-op parity tests */
void SUF(osvos_oracle_conv_fwd)(const REAL* x, const REAL* w, const REAL* b, REAL* y, int N, int Cin, int H, int W, int Cout, int K, int relu) {
  SUF(conv_fwd)(x, w, b, y, N, Cin, H, W, Cout, K, relu);
}
void SUF(osvos_oracle_conv_bwd)(const REAL* x, const REAL* w, const REAL* dy, REAL* dx, REAL* dw, REAL* db, int N, int Cin, int H, int W, int Cout, int K) {
  SUF(conv_bwd)(x, w, dy, dx, dw, db, N, Cin, H, W, Cout, K);
}
void SUF(osvos_oracle_pool_fwd)(const REAL* x, REAL* y, int* arg, int N, int C, int H, int W) { SUF(pool_fwd)(x, y, arg, N, C, H, W); }
void SUF(osvos_oracle_pool_bwd_acc)(const REAL* dy, const int* arg, REAL* dx, int N, int C, int H, int W) { SUF(pool_bwd_acc)(dy, arg, dx, N, C, H, W); }
void SUF(osvos_oracle_deconv_fwd)(const REAL* x, const REAL* w, REAL* y, int N, int Cin, int Cout, int h, int wd, int k, int s) { SUF(deconv_fwd)(x, w, y, N, Cin, Cout, h, wd, k, s); }
void SUF(osvos_oracle_deconv_bwd_in)(const REAL* dy, const REAL* w, REAL* dx, int N, int Cin, int Cout, int h, int wd, int k, int s) { SUF(deconv_bwd_in)(dy, w, dx, N, Cin, Cout, h, wd, k, s); }
