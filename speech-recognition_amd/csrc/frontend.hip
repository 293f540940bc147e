// Fused audio front end for gfx950: framing + Hann window + DFT (f32 MFMA) + |.|^2 + mel filterbank
// + log + SpecAugment masks + delta / delta-delta + zero padding, one pass over the audio.
// Replaces data.py:169-187, 282-301, 319-324 and the padded_batch of run/train.py:189-197.
// The same kernel serves the other two feature types of data_config.py:77-101: "spectrogram" (data.py:122-142:
// |STFT|, the mel stage is replaced by a square root) and "mfcc" (data.py:192-241: log-mel, then the DCT-II of
// tf.signal.mfccs_from_log_mel_spectrograms as a small table product, first num_mfcc coefficients).
//
// One workgroup = one clip x 16*MT consecutive frames (MT = 4 unless the frame parameters need more LDS
// than a CU has; all but 2 are written, 2 are halo frames for the two causal differences).  HBM traffic is the algorithmic minimum: every sample is read once per tile
// (+2 halo frames) and every output element written once; the twiddle table (window folded in,
// stored in MFMA fragment order) and the mel matrix stay L2 resident.
#include <math.h>
#include <stdlib.h>

#include <vector>

#include "common.h"

#define FE_FR_MAX 64  // frames per workgroup (incl. 2 halo) at MT = 4
#define FE_MAXBAND 16

struct FeArgs {
  const float* audio;
  const int32_t* n_samples;
  const float* tw;
  const float* melw;
  const int32_t* melrange;
  const uint32_t* seed;
  float* out;
  int B, n_max, T_out;
  int L, step, bins, nmel, C;
  int KS, NBT, pad, seg_len, seg_floats, PLD;
  int sym, KS2;                 // sym: fft_length == frame_length (even): cos/sin symmetry folds the frame, KS2 = DFT k-steps
  int mode, nf, lmw;            // feature type (0 log-mel, 1 spectrogram, 2 mfcc), features per frame, row width of Lm
  const float* dct;             // mfcc: [nmel][nf] DCT-II table
  float eps;
  int sa_enable, sa_F, sa_mF, sa_T, sa_mT;
  float sa_p;
};

template <int MT>
__global__ __launch_bounds__(256) void logmel_kernel(FeArgs a) {
  constexpr int FE_FR = 16 * MT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* seg = smem;                         // padded sample segment
  float* P = seg + a.seg_floats;             // [FE_FR][PLD] power spectrum
  float* Lm = P + FE_FR * a.PLD;             // [FE_FR][lmw] masked features (mfcc: the unmasked log-mel)
  float* Fm = Lm + FE_FR * a.lmw;            // mfcc only: [FE_FR][nf] masked cepstral coefficients
  int* bands = reinterpret_cast<int*>(Fm + (a.mode == 2 ? FE_FR * a.nf : 0));  // [4][FE_MAXBAND]: f0, f, t0, t

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int t0 = (int)blockIdx.x * (FE_FR - 2) - 2;  // first computed frame (may be negative)
  const int n_b = a.n_samples[b];
  const int T_b = n_b >= a.L ? 1 + (n_b - a.L) / a.step : 0;
  const int fstride = a.step + a.pad;

  // SpecAugment draws for this clip (data.py:282-301), one thread, sequential like the reference
  if (tid == 0) {
    for (int i = 0; i < FE_MAXBAND; ++i) { bands[i] = 0; bands[FE_MAXBAND + i] = 0; bands[2 * FE_MAXBAND + i] = 0; bands[3 * FE_MAXBAND + i] = 0; }
    if (a.sa_enable) {
      const AsrRngKey key = asr_rng_key(a.seed[0], 3u /* STREAM_SPECAUG */);
      if (a.sa_F > 0 && a.sa_mF > 0) {
        for (int i = 0; i < a.sa_mF; ++i) {
          const int f = asr_uniform_int(key, (uint32_t)(b * 64 + 2 * i), a.sa_F);
          const int f0 = asr_uniform_int(key, (uint32_t)(b * 64 + 2 * i + 1), a.nf - f);
          bands[i] = f0; bands[FE_MAXBAND + i] = f;
        }
      }
      if (a.sa_T > 0 && a.sa_mT > 0 && a.sa_p > 0.f) {
        int applied = 0;
        const int max_maskable = (int)((float)T_b * a.sa_p);
        for (int j = 0; j < a.sa_mT; ++j) {
          int t = asr_uniform_int(key, (uint32_t)(b * 64 + 32 + 2 * j), a.sa_T);
          t = min(t, max_maskable - applied);
          t = max(t, 0);
          applied += t;
          const int tt0 = asr_uniform_int(key, (uint32_t)(b * 64 + 32 + 2 * j + 1), T_b - t);
          bands[2 * FE_MAXBAND + j] = tt0; bands[3 * FE_MAXBAND + j] = t;
        }
      }
    }
  }

  // a. stage the sample segment (coalesced), padded so that frame rows start 2 banks apart
  const long g0 = (long)t0 * a.step;
  const float* clip = a.audio + (long)b * a.n_max;
  for (int p = tid; p < a.seg_len; p += 256) {
    const long g = g0 + p;
    const float v = (g >= 0 && g < n_b) ? clip[g] : 0.f;
    seg[p + a.pad * (p / a.step)] = v;
  }
  __syncthreads();

  // b. windowed DFT on the MFMA: [64 frames x L] x [L x (cos|sin) 16 bins], power -> P
  const int li = lane & 15, lq = lane >> 4;
  for (int bt = wave; bt < a.NBT; bt += 4) {
    f32x4 ac[MT], as[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) { ac[m] = (f32x4){0.f, 0.f, 0.f, 0.f}; as[m] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const float* twc = a.tw + (long)(bt * 2 + 0) * a.KS2 * 64 + lane;
    const float* tws = a.tw + (long)(bt * 2 + 1) * a.KS2 * 64 + lane;
    if (a.sym) {
      // cos(2 pi k (L-n)/L) = cos(2 pi k n/L), sin(...) = -sin(...), and the periodic Hann window is symmetric
      // too (w[n] = w[L-n], w[0] = 0): pair samples n and L-n, n = 1 .. L/2 -> half the MFMAs.
      //   Re[k] = sum_n (x[n] + x[L-n]) * w[n] cos(2 pi k n / L)   (the n = L/2 entry of the table is halved)
      //   Im[k] = sum_n (x[n] - x[L-n]) * w[n] sin(2 pi k n / L)
      for (int ks = 0; ks < a.KS2; ++ks) {
        const float bc = twc[ks * 64], bs = tws[ks * 64];
        const int n = min(4 * ks + lq + 1, a.L / 2), n2 = a.L - n;   // entries past L/2 carry zero twiddles
        const int o1 = n + a.pad * (n / a.step), o2 = n2 + a.pad * (n2 / a.step);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const float* row = seg + (m * 16 + li) * fstride;
          const float x1 = row[o1], x2 = row[o2];
          ac[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1 + x2, bc, ac[m], 0, 0, 0);
          as[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1 - x2, bs, as[m], 0, 0, 0);
        }
      }
    } else {
      for (int ks = 0; ks < a.KS2; ++ks) {
        const float bc = twc[ks * 64], bs = tws[ks * 64];
        const int k = 4 * ks + lq;
        const int koff = k + a.pad * (k / a.step);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const float av = seg[(m * 16 + li) * fstride + koff];
          ac[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bc, ac[m], 0, 0, 0);
          as[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bs, as[m], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int frame = m * 16 + lq * 4 + r;
        P[frame * a.PLD + bt * 16 + li] = ac[m][r] * ac[m][r] + as[m][r] * as[m][r];
      }
  }
  __syncthreads();

  // c. mel filterbank (triangles are sparse: only bins melrange[m]..melrange[nmel+m]) + log + masks
  //    (spectrogram: magnitude instead; mfcc: unmasked log-mel here, DCT + masks in c2)
  const int ncol = a.mode == 1 ? a.nf : a.nmel;
  for (int idx = tid; idx < FE_FR * ncol; idx += 256) {
    const int frame = idx / ncol, m = idx - frame * ncol;
    const int t = t0 + frame;
    float v = 0.f;
    if (t >= 0 && t < T_b) {
      if (a.mode == 1) {
        v = sqrtf(P[frame * a.PLD + m]);                         // tf.abs(stft)
      } else {
        const int lo = a.melrange[m], hi = a.melrange[a.nmel + m];
        float s = 0.f;
        for (int bin = lo; bin <= hi; ++bin) s = fmaf(P[frame * a.PLD + bin], a.melw[bin * a.nmel + m], s);
        // f32 add like tf.math.log(mel + eps), then a correctly rounded log (the reference's silence
        // fixture pins log(1e-12f) to the last bit; the fast f32 log is 1 ulp off there)
        const float se = s + a.eps;
        v = (float)log((double)se);
      }
      if (a.mode != 2) {
        bool zero = false;
        for (int i = 0; i < a.sa_mF && i < FE_MAXBAND; ++i)
          zero |= (m >= bands[i] && m < bands[i] + bands[FE_MAXBAND + i]);
        for (int j = 0; j < a.sa_mT && j < FE_MAXBAND; ++j)
          zero |= (t >= bands[2 * FE_MAXBAND + j] && t < bands[2 * FE_MAXBAND + j] + bands[3 * FE_MAXBAND + j]);
        if (a.sa_enable && zero) v = 0.f;
      }
    }
    Lm[frame * a.lmw + m] = v;
  }
  __syncthreads();
  const float* feat = Lm;
  int fw = a.lmw;
  if (a.mode == 2) {
    // c2. mfcc[k] = sum_m logmel[m] * 2 cos(pi k (2m + 1) / (2 nmel)) / sqrt(2 nmel)   (data.py:233-234)
    for (int idx = tid; idx < FE_FR * a.nf; idx += 256) {
      const int frame = idx / a.nf, k = idx - frame * a.nf;
      const int t = t0 + frame;
      float v = 0.f;
      if (t >= 0 && t < T_b) {
        for (int m = 0; m < a.nmel; ++m) v = fmaf(Lm[frame * a.lmw + m], a.dct[m * a.nf + k], v);
        bool zero = false;
        for (int i = 0; i < a.sa_mF && i < FE_MAXBAND; ++i)
          zero |= (k >= bands[i] && k < bands[i] + bands[FE_MAXBAND + i]);
        for (int j = 0; j < a.sa_mT && j < FE_MAXBAND; ++j)
          zero |= (t >= bands[2 * FE_MAXBAND + j] && t < bands[2 * FE_MAXBAND + j] + bands[3 * FE_MAXBAND + j]);
        if (a.sa_enable && zero) v = 0.f;
      }
      Fm[idx] = v;
    }
    __syncthreads();
    feat = Fm;
    fw = a.nf;
  }

  // d. delta / delta-delta (same rounding order as data.py:319-321) and coalesced store
  const int per_frame = a.nf * a.C;
  for (int idx = tid; idx < (FE_FR - 2) * per_frame; idx += 256) {
    const int fo = idx / per_frame + 2, rem = idx % per_frame;
    const int m = rem / a.C, c = rem - m * a.C;
    const int t = t0 + fo;
    if (t >= a.T_out) continue;
    float v = 0.f;
    if (t < T_b) {
      const float x0 = feat[fo * fw + m], x1 = feat[(fo - 1) * fw + m], x2 = feat[(fo - 2) * fw + m];
      const float d0 = x0 - x1, d1 = x1 - x2;
      v = c == 0 ? x0 : (c == 1 ? d0 : d0 - d1);
    }
    a.out[((long)b * a.T_out + t) * per_frame + rem] = v;
  }
}

// ------------------------------------------------------------------------------------------ host
static int fe_geometry(const asr_logmel_cfg* c, FeArgs* a, int FE_FR = FE_FR_MAX) {
  ASR_CHECK(c->frame_length > 0 && c->frame_step > 0 && c->fft_length > 0, ASR_ERR_SHAPE,
            "logmel: need frame_length>0, frame_step>0, fft_length>0");
  ASR_CHECK(c->feature_type >= 0 && c->feature_type <= 2, ASR_ERR_ARG, "logmel: feature_type %d (0 log-mel, 1 spectrogram, 2 mfcc)", c->feature_type);
  ASR_CHECK(c->feature_type == 1 || c->num_mel_bins > 0, ASR_ERR_SHAPE, "logmel: need num_mel_bins>0");
  ASR_CHECK(c->feature_type != 2 || (c->num_mfcc > 0 && c->num_mfcc <= c->num_mel_bins), ASR_ERR_SHAPE,
            "logmel: need 0 < num_mfcc <= num_mel_bins, got %d / %d", c->num_mfcc, c->num_mel_bins);
  ASR_CHECK(c->sa_mF <= FE_MAXBAND && c->sa_mT <= FE_MAXBAND, ASR_ERR_SHAPE, "logmel: m_F/m_T > %d", FE_MAXBAND);
  a->L = c->frame_length; a->step = c->frame_step; a->bins = c->fft_length / 2 + 1;
  a->mode = c->feature_type;
  a->nmel = a->mode == 1 ? 1 : c->num_mel_bins;                 // spectrogram: the mel tables are one dummy column
  a->nf = a->mode == 0 ? a->nmel : (a->mode == 1 ? a->bins : c->num_mfcc);
  a->lmw = a->mode == 1 ? a->nf : a->nmel;
  a->C = c->use_delta ? 3 : 1;
  a->KS = asr_cdiv(a->L, 4); a->NBT = asr_cdiv(a->bins, 16);
  a->sym = (c->fft_length == c->frame_length && c->frame_length % 2 == 0) ? 1 : 0;
  a->KS2 = a->sym ? asr_cdiv(a->L / 2, 4) : a->KS;
  a->pad = ((2 - a->step) % 32 + 32) % 32;
  a->seg_len = (FE_FR - 1) * a->step + 4 * a->KS;
  a->seg_floats = a->seg_len + a->pad * (a->seg_len / a->step + 1);
  a->seg_floats = (a->seg_floats + 3) & ~3;
  a->PLD = a->NBT * 16 + 4;
  a->eps = c->epsilon;
  a->sa_enable = c->sa_enable; a->sa_F = c->sa_F; a->sa_mF = c->sa_enable ? c->sa_mF : 0; a->sa_T = c->sa_T;
  a->sa_mT = c->sa_enable ? c->sa_mT : 0; a->sa_p = c->sa_p;
  return ASR_OK;
}
static size_t fe_smem_bytes(const FeArgs& a, int FE_FR) {
  return sizeof(float) * ((size_t)a.seg_floats + (size_t)FE_FR * a.PLD + (size_t)FE_FR * a.lmw + (a.mode == 2 ? (size_t)FE_FR * a.nf : 0)) +
         sizeof(int) * 4 * FE_MAXBAND;
}

extern "C" int asr_logmel_table_sizes(const asr_logmel_cfg* cfg, long* n_tw, long* n_melw, long* n_range) {
  ASR_CHECK(cfg && n_tw && n_melw && n_range, ASR_ERR_ARG, "asr_logmel_table_sizes: null argument");
  FeArgs a{};
  int rc = fe_geometry(cfg, &a);
  if (rc) return rc;
  *n_tw = (long)a.NBT * 2 * a.KS2 * 64;
  *n_melw = (long)a.bins * a.nmel + (a.mode == 2 ? (long)a.nmel * a.nf : 0);    // mfcc: the DCT table follows the mel matrix
  *n_range = 2L * a.nmel;
  return ASR_OK;
}

extern "C" int asr_logmel_build_tables(const asr_logmel_cfg* cfg, float* tw, float* melw, int32_t* range) {
  ASR_CHECK(cfg && tw && melw && range, ASR_ERR_ARG, "asr_logmel_build_tables: null argument");
  FeArgs a{};
  int rc = fe_geometry(cfg, &a);
  if (rc) return rc;
  const double PI = 3.14159265358979323846;
  // twiddles with the periodic Hann window folded in ([TF-sem] tf.signal.hann_window(periodic=True))
  for (int bt = 0; bt < a.NBT; ++bt)
    for (int cs = 0; cs < 2; ++cs)
      for (int ks = 0; ks < a.KS2; ++ks)
        for (int lane = 0; lane < 64; ++lane) {
          const int bin = bt * 16 + (lane & 15);
          double v = 0.0;
          if (a.sym) {
            const int n = 4 * ks + (lane >> 4) + 1;        // sample pair (n, L - n), n = 1 .. L/2
            if (n <= a.L / 2 && bin < a.bins) {
              const double w = 0.5 - 0.5 * cos(2.0 * PI * n / a.L);
              const long kb = ((long)n * bin) % a.L;
              const double ang = 2.0 * PI * (double)kb / a.L;
              v = w * (cs == 0 ? cos(ang) : sin(ang));
              if (n == a.L / 2) v = cs == 0 ? 0.5 * v : 0.0;   // its own partner: x + x = 2x, x - x = 0
            }
          } else {
            const int k = 4 * ks + (lane >> 4);
            // [TF-sem] tf.signal.stft: rfft crops the windowed frame to fft_length samples when it is shorter
            if (k < a.L && k < cfg->fft_length && bin < a.bins) {
              const double w = 0.5 - 0.5 * cos(2.0 * PI * k / a.L);
              const long kb = ((long)k * bin) % cfg->fft_length;
              const double ang = 2.0 * PI * (double)kb / cfg->fft_length;
              v = w * (cs == 0 ? cos(ang) : sin(ang));
            }
          }
          tw[((long)(bt * 2 + cs) * a.KS2 + ks) * 64 + lane] = (float)v;
        }
  if (a.mode == 1) {                                             // spectrogram: no mel stage, one dummy column
    for (int bin = 0; bin < a.bins; ++bin) melw[bin] = 0.f;
    range[0] = 0; range[1] = -1;
    return ASR_OK;
  }
  if (a.mode == 2) {
    // [TF-sem] tf.signal.mfccs_from_log_mel_spectrograms = dct(type 2, unnormalised: 2 sum x cos(pi k (2m+1) / 2N)) * rsqrt(2N)
    float* dct = melw + (long)a.bins * a.nmel;
    const double sc = 2.0 / sqrt(2.0 * a.nmel);
    for (int m = 0; m < a.nmel; ++m)
      for (int k = 0; k < a.nf; ++k) dct[(long)m * a.nf + k] = (float)(sc * cos(PI * k * (2.0 * m + 1.0) / (2.0 * a.nmel)));
  }
  // [TF-sem] tf.signal.linear_to_mel_weight_matrix: HTK mel, triangles on the mel axis, DC row zero
  auto hz2mel = [](double f) { return 1127.0 * log1p(f / 700.0); };
  const double nyq = cfg->sample_rate / 2.0;
  const double mlo = hz2mel(cfg->lower_edge_hertz), mhi = hz2mel(cfg->upper_edge_hertz);
  std::vector<double> edges(a.nmel + 2);
  for (int i = 0; i < a.nmel + 2; ++i) edges[i] = mlo + (mhi - mlo) * i / (a.nmel + 1);
  for (int m = 0; m < a.nmel; ++m) { range[m] = a.bins; range[a.nmel + m] = -1; }
  for (int bin = 0; bin < a.bins; ++bin) {
    const double mel = hz2mel(nyq * bin / (a.bins - 1));
    for (int m = 0; m < a.nmel; ++m) {
      double w = 0.0;
      if (bin > 0) {
        const double lo = (mel - edges[m]) / (edges[m + 1] - edges[m]);
        const double up = (edges[m + 2] - mel) / (edges[m + 2] - edges[m + 1]);
        w = fmax(0.0, fmin(lo, up));
      }
      const float wf = (float)w;
      melw[(long)bin * a.nmel + m] = wf;
      if (wf != 0.f) {
        if (bin < range[m]) range[m] = bin;
        if (bin > range[a.nmel + m]) range[a.nmel + m] = bin;
      }
    }
  }
  for (int m = 0; m < a.nmel; ++m)
    if (range[a.nmel + m] < range[m]) { range[m] = 0; range[a.nmel + m] = -1; }  // empty filter
  return ASR_OK;
}

extern "C" int asr_logmel_features(const asr_logmel_cfg* cfg, const float* audio, const int32_t* n_samples, int B,
                                   int n_max, const float* tw, const float* melw, const int32_t* melrange,
                                   const uint32_t* seed, float* out, int T_out, void* stream) {
  ASR_CHECK(cfg && audio && n_samples && tw && melw && melrange && out, ASR_ERR_ARG, "asr_logmel_features: null argument");
  ASR_CHECK(B > 0 && n_max > 0 && T_out > 0, ASR_ERR_SHAPE, "asr_logmel_features: B, n_max, T_out must be > 0");
  ASR_CHECK(!(cfg->sa_enable && !seed), ASR_ERR_ARG, "asr_logmel_features: SpecAugment needs a device seed");
  FeArgs a{};
  int MT = 4, rc = 0;
  size_t smem = 0;
  // Frames per workgroup: the kernel's phases (stage, DFT, mel + log, delta + store) run one after another
  // inside a workgroup, so several workgroups per CU (<= 40 KiB of LDS each: measured 170 us at 16 frames per
  // workgroup vs 194 at 32 and 312 at 64 for the libri configuration) overlap them; frame parameters too long
  // for that fall back to whatever fits the 160 KiB of a CU.
  const int env_mt = getenv("ASR_LOGMEL_MT") ? atoi(getenv("ASR_LOGMEL_MT")) : 0;   // tuning aid: 4, 2 or 1
  for (int pass = 0; pass < 2; ++pass) {
    const size_t limit = pass == 0 ? 40 * 1024 : 160 * 1024;
    for (MT = env_mt ? env_mt : 4; MT >= 1; MT >>= 1) {
      rc = fe_geometry(cfg, &a, 16 * MT);
      if (rc) return rc;
      smem = fe_smem_bytes(a, 16 * MT);
      if (smem <= limit || (env_mt && smem <= 160 * 1024)) break;
    }
    if (MT >= 1) break;
  }
  ASR_CHECK(MT >= 1, ASR_ERR_SHAPE, "asr_logmel_features: frame parameters need %zu B of LDS (> 160 KiB)", smem);
  a.audio = audio; a.n_samples = n_samples; a.tw = tw; a.melw = melw; a.melrange = melrange; a.seed = seed; a.out = out;
  a.dct = melw + (long)a.bins * a.nmel;
  a.B = B; a.n_max = n_max; a.T_out = T_out;
  static unsigned long long attr_set = 0;
  if (asr_first_use_on_device(attr_set)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(logmel_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(logmel_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(logmel_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  dim3 grid((unsigned)asr_cdiv(T_out, 16 * MT - 2), (unsigned)B);
  if (MT == 4) hipLaunchKernelGGL(logmel_kernel<4>, grid, dim3(256), smem, (hipStream_t)stream, a);
  else if (MT == 2) hipLaunchKernelGGL(logmel_kernel<2>, grid, dim3(256), smem, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(logmel_kernel<1>, grid, dim3(256), smem, (hipStream_t)stream, a);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
