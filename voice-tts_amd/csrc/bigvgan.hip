// bigvgan.hip -- BigVGAN-v2 generator forward on gfx950: host orchestration + C ABI (seam 2).
//
// Reference: indextts/s2mel/modules/bigvgan/bigvgan.py:243-400 (BigVGAN), :31-147 (AMPBlock1);
// call site indextts/infer_v2.py:154-158,735.
//
// HBM layout
//   weight arena (one hipMalloc, broadcastable): per conv the weights as three bf16 planes
//   Wx[phase*tap][Cin_pad/16][plane][2][Cout_pad][8] (conv1d_x3.hip; IXTTS_BV_CONV=f32: Wq[phase][tap][Cin_pad/8][2][Cout_pad][4]
//   fp32 for the fp32-MFMA kernel of conv1d.hip) + bias[Cout]; per activation log_alpha[C], log_beta[C]; the 12 filter taps.
//   activations: 11 buffers of 6144*F*B floats ([B][C][T] row-major, T contiguous):
//     XS  previous stage output      X   stage input (after ups)
//     per resblock branch j (3 of them): R_j running state / branch result, T1_j activation output, T2_j conv1 output
//   conv inputs (x3 mode): 4 plane buffers [B][3][C/8][T][8] bf16 -- one per branch (written by the Snake passes) and one for
//   the mel / the stage outputs the transposed convs read (written by the split pass)
//
// The three AMPBlock1 branches of a stage (k = 3, 7, 11) read the same input and only meet in the mean, so they run
// CONCURRENTLY on three streams (forked after the up-sampling conv, joined in front of the k = 11 branch's last conv,
// whose epilogue forms ((r0 + r1) + r2) / 3): the conv tiles of one branch fill the CUs another branch's last partial
// round of tiles leaves idle, and the HBM-bound activation passes run beside MFMA-bound convs.
#include <map>
#include <vector>

#include "conv.h"

namespace ixtts {

struct ConvDesc {
  size_t w_off = 0, b_off = 0;  // float offsets into the arena
  int Cin = 0, Cout = 0, Cin_pad = 0, Cout_pad = 0, K = 0, dil = 1, pad = 0;
  int stride = 1;  // >1: transposed conv
  bool has_bias = true;
  bool w_set = false, b_set = false;
};

struct ActDesc {
  size_t a_off = 0, b_off = 0;
  int C = 0;
  bool a_set = false, b_set = false;
};

}  // namespace ixtts

using namespace ixtts;

struct ixtts_bigvgan {
  ixtts_bigvgan_cfg cfg;
  std::map<std::string, ConvDesc> convs;
  std::map<std::string, ActDesc> acts;
  float* arena = nullptr;
  size_t arena_floats = 0;
  size_t filt_off = 0, zero_off = 0;
  static constexpr int NBUF = 11;
  float* buf[NBUF] = {};
  size_t buf_floats = 0;
  bool x3 = true;      // convs on the bf16 matrix cores with three-way split operands (conv1d_x3.hip); IXTTS_BV_CONV=f32: conv1d.hip
  void* pbuf[4] = {};  // x planes: one per resblock branch + one for the mel / stage outputs
  size_t pbuf_bytes = 0;
  hipStream_t side[2] = {nullptr, nullptr};  // streams of the k = 3 / k = 7 branches (the caller's stream carries k = 11)
  hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
  bool concurrent = true;                    // IXTTS_BV_STREAMS=1 keeps everything on the caller's stream
  int total_up = 1;
  bool finalized = false;
};

static int round_up(int x, int m) { return (x + m - 1) / m * m; }

static void add_conv(ixtts_bigvgan* h, const std::string& name, int Cin, int Cout, int K, int dil, int stride, bool bias,
                     size_t& off) {
  ConvDesc d;
  d.Cin = Cin;
  d.Cout = Cout;
  d.K = K;
  d.dil = dil;
  d.stride = stride;
  d.has_bias = bias;
  d.Cin_pad = round_up(Cin, h->x3 ? 16 : 8);
  d.Cout_pad = h->x3 ? conv_x3_cout_pad(Cout) : round_up(Cout, conv_tile_bm(Cout));
  d.pad = stride == 1 ? (K * dil - dil) / 2 : (K - stride) / 2;
  d.w_off = off;
  // K taps in total (phases * taps-per-phase for transposed); three bf16 planes = 6 bytes per weight in x3 mode
  off += h->x3 ? (size_t)K * d.Cin_pad * d.Cout_pad * 3 / 2 : (size_t)K * d.Cin_pad * d.Cout_pad;
  off = (off + 3) & ~(size_t)3;
  d.b_off = off;
  off += round_up(Cout, 4);
  h->convs[name] = d;
}

static void add_act(ixtts_bigvgan* h, const std::string& name, int C, size_t& off) {
  ActDesc a;
  a.C = C;
  a.a_off = off;
  off += round_up(C, 4);
  a.b_off = off;
  off += round_up(C, 4);
  h->acts[name] = a;
}

extern "C" int ixtts_bigvgan_create(ixtts_bigvgan** out, const ixtts_bigvgan_cfg* cfg) {
  IX_ARG(out && cfg, "bigvgan_create: null argument");
  IX_ARG(cfg->n_stages >= 1 && cfg->n_stages <= IXTTS_BIGVGAN_MAX_STAGES, "bigvgan_create: n_stages %d", cfg->n_stages);
  IX_ARG(cfg->n_resblock_kernels >= 1 && cfg->n_resblock_kernels <= IXTTS_BIGVGAN_MAX_RESK, "bigvgan_create: n_resblock_kernels");
  IX_ARG(cfg->n_resblock_kernels == 3, "bigvgan_create: the fused /3 epilogue assumes 3 resblocks per stage");
  IX_ARG(cfg->num_mels > 0 && cfg->upsample_initial_channel > 0 && cfg->max_frames > 0, "bigvgan_create: bad sizes");
  IX_ARG((cfg->upsample_initial_channel >> cfg->n_stages) >= 1, "bigvgan_create: channels vanish");
  auto* h = new (std::nothrow) ixtts_bigvgan();
  if (!h) return IXTTS_ERR_NOMEM;
  h->cfg = *cfg;
  if (const char* e = getenv("IXTTS_BV_STREAMS")) h->concurrent = strcmp(e, "1") != 0;
  if (const char* e = getenv("IXTTS_BV_CONV")) h->x3 = strcmp(e, "f32") != 0;
  size_t off = 0;
  h->filt_off = off;
  off += 16;
  int c = cfg->upsample_initial_channel;
  add_conv(h, "conv_pre", cfg->num_mels, c, 7, 1, 1, true, off);
  size_t max_ct = (size_t)c;  // C*T per frame
  int up = 1;
  for (int i = 0; i < cfg->n_stages; ++i) {
    int u = cfg->upsample_rates[i], ku = cfg->upsample_kernel_sizes[i];
    if (u < 1 || ku % u != 0 || (ku - u) % 2 != 0) {
      delete h;
      set_error("bigvgan_create: stage %d: kernel %d must be a multiple of stride %d with even k-s", i, ku, u);
      return IXTTS_ERR_ARG;
    }
    add_conv(h, "ups." + std::to_string(i) + ".0", c, c / 2, ku, 1, u, true, off);
    c /= 2;
    up *= u;
    if ((size_t)c * up > max_ct) max_ct = (size_t)c * up;
    for (int j = 0; j < cfg->n_resblock_kernels; ++j) {
      std::string p = "resblocks." + std::to_string(i * cfg->n_resblock_kernels + j);
      int k = cfg->resblock_kernel_sizes[j];
      for (int m = 0; m < 3; ++m) {
        add_conv(h, p + ".convs1." + std::to_string(m), c, c, k, cfg->resblock_dilations[j][m], 1, true, off);
        add_conv(h, p + ".convs2." + std::to_string(m), c, c, k, 1, 1, true, off);
      }
      for (int m = 0; m < 6; ++m) add_act(h, p + ".activations." + std::to_string(m) + ".act", c, off);
    }
  }
  add_act(h, "activation_post.act", c, off);
  // conv_post is Cout=1: stored unpacked [C][7]
  {
    ConvDesc d;
    d.Cin = c;
    d.Cout = 1;
    d.K = 7;
    d.has_bias = false;
    d.b_set = true;
    d.w_off = off;
    off += round_up(c * 7, 4);
    d.b_off = off;
    off += 4;
    h->convs["conv_post"] = d;
  }
  h->zero_off = off;  // 64 floats that stay zero: where the conv kernel points loads it must not make (ConvParams::zeros)
  off += 64;
  h->total_up = up;
  h->arena_floats = off;
  if (hipMalloc(&h->arena, off * sizeof(float)) != hipSuccess) {
    delete h;
    set_error("bigvgan_create: hipMalloc(%zu) failed", off * sizeof(float));
    return IXTTS_ERR_NOMEM;
  }
  hipMemset(h->arena, 0, off * sizeof(float));
  *out = h;
  return IXTTS_OK;
}

static int ensure_workspace(ixtts_bigvgan* h, int B, int F) {
  size_t per_frame = 0;
  int c = h->cfg.upsample_initial_channel, up = 1;
  per_frame = (size_t)c;
  for (int i = 0; i < h->cfg.n_stages; ++i) {
    c /= 2;
    up *= h->cfg.upsample_rates[i];
    if ((size_t)c * up > per_frame) per_frame = (size_t)c * up;
  }
  size_t need = per_frame * (size_t)F * B;
  if (need <= h->buf_floats) return IXTTS_OK;
  for (int i = 0; i < ixtts_bigvgan::NBUF; ++i) {
    if (h->buf[i]) hipFree(h->buf[i]);
    h->buf[i] = nullptr;
  }
  h->buf_floats = 0;
  for (int i = 0; i < ixtts_bigvgan::NBUF; ++i) {
    if (hipMalloc(&h->buf[i], need * sizeof(float)) != hipSuccess) {
      set_error("bigvgan: workspace hipMalloc(%zu) failed", need * sizeof(float));
      return IXTTS_ERR_NOMEM;
    }
  }
  h->buf_floats = need;
  if (h->x3) {
    // a plane buffer holds the largest conv input: C rounded up to 8 channels x T x 3 planes x 2 bytes
    size_t pe = (size_t)round_up(h->cfg.num_mels, 8);
    int c2 = h->cfg.upsample_initial_channel, up2 = 1;
    if ((size_t)c2 > pe) pe = c2;
    for (int i = 0; i < h->cfg.n_stages; ++i) {
      c2 /= 2;
      up2 *= h->cfg.upsample_rates[i];
      if ((size_t)round_up(c2, 8) * up2 > pe) pe = (size_t)round_up(c2, 8) * up2;
    }
    const size_t pneed = pe * (size_t)F * B * 6;
    for (int i = 0; i < 4; ++i) {
      if (h->pbuf[i]) hipFree(h->pbuf[i]);
      h->pbuf[i] = nullptr;
      if (hipMalloc(&h->pbuf[i], pneed) != hipSuccess) {
        set_error("bigvgan: plane workspace hipMalloc(%zu) failed", pneed);
        return IXTTS_ERR_NOMEM;
      }
    }
    h->pbuf_bytes = pneed;
  }
  return IXTTS_OK;
}

extern "C" int ixtts_bigvgan_set_tensor(ixtts_bigvgan* h, const char* name, const float* data, const int64_t* shape,
                                        int ndim) {
  IX_ARG(h && name && data && shape, "bigvgan_set_tensor: null argument");
  std::string n(name);
  auto ends = [&](const char* s) { size_t L = strlen(s); return n.size() >= L && n.compare(n.size() - L, L, s) == 0; };
  if (ends(".alpha") || ends(".beta")) {
    bool is_a = ends(".alpha");
    std::string base = n.substr(0, n.rfind('.'));
    auto it = h->acts.find(base);
    if (it == h->acts.end()) { set_error("bigvgan_set_tensor: unknown tensor '%s'", name); return IXTTS_ERR_NAME; }
    ActDesc& a = it->second;
    IX_ARG(ndim == 1 && shape[0] == a.C, "bigvgan_set_tensor: %s expects [%d]", name, a.C);
    IX_HIP(hipMemcpy(h->arena + (is_a ? a.a_off : a.b_off), data, a.C * sizeof(float), hipMemcpyHostToDevice));
    (is_a ? a.a_set : a.b_set) = true;
    return IXTTS_OK;
  }
  bool is_w = ends(".weight"), is_b = ends(".bias");
  if (!is_w && !is_b) { set_error("bigvgan_set_tensor: unknown tensor '%s'", name); return IXTTS_ERR_NAME; }
  std::string base = n.substr(0, n.rfind('.'));
  auto it = h->convs.find(base);
  if (it == h->convs.end()) { set_error("bigvgan_set_tensor: unknown tensor '%s'", name); return IXTTS_ERR_NAME; }
  ConvDesc& d = it->second;
  if (is_b) {
    IX_ARG(d.has_bias, "bigvgan_set_tensor: %s has no bias in this configuration", base.c_str());
    IX_ARG(ndim == 1 && shape[0] == d.Cout, "bigvgan_set_tensor: %s expects [%d]", name, d.Cout);
    IX_HIP(hipMemcpy(h->arena + d.b_off, data, d.Cout * sizeof(float), hipMemcpyHostToDevice));
    d.b_set = true;
    return IXTTS_OK;
  }
  IX_ARG(ndim == 3, "bigvgan_set_tensor: %s expects 3 dims", name);
  if (base == "conv_post") {
    IX_ARG(shape[0] == 1 && shape[1] == d.Cin && shape[2] == 7, "bigvgan_set_tensor: conv_post.weight expects [1,%d,7]", d.Cin);
    IX_HIP(hipMemcpy(h->arena + d.w_off, data, d.Cin * 7 * sizeof(float), hipMemcpyHostToDevice));
    d.w_set = true;
    return IXTTS_OK;
  }
  // w(tapslot, ci, co) goes to
  //   x3:  plane p of Wx[tapslot][ci/16][p][(ci%16)/8][co][ci%8] (bf16)                          (conv1d_x3.hip)
  //   f32: Wq[tapslot][ci/8][ci%2][co][(ci%8)/2]                                                   (conv1d.hip)
  const size_t nw = (size_t)d.K * d.Cin_pad * d.Cout_pad;
  std::vector<float> packed(h->x3 ? nw * 3 / 2 : nw, 0.f);
  uint16_t* planes = reinterpret_cast<uint16_t*>(packed.data());
  const int ngroups = d.Cin_pad / 8, ng16 = d.Cin_pad / 16;
  auto put = [&](int tapslot, int ci, int co, float v) {
    if (!h->x3) {
      const int g = ci / 8, r = ci % 8;
      packed[((((size_t)tapslot * ngroups + g) * 2 + (r & 1)) * d.Cout_pad + co) * 4 + (r >> 1)] = v;
      return;
    }
    // v = h + m + l exactly, each piece the remainder before it rounded to 8 significant bits (ties away)
    uint32_t u, ru, lu;
    memcpy(&u, &v, 4);
    const uint32_t hu = (u + 0x8000u) & 0xffff0000u;
    float hf, mf;
    memcpy(&hf, &hu, 4);
    const float r = v - hf;
    memcpy(&ru, &r, 4);
    const uint32_t mu = (ru + 0x8000u) & 0xffff0000u;
    memcpy(&mf, &mu, 4);
    const float lf = r - mf;
    memcpy(&lu, &lf, 4);
    const uint32_t piece[3] = {hu, mu, lu};
    for (int pl = 0; pl < 3; ++pl)
      planes[(((((size_t)tapslot * ng16 + ci / 16) * 3 + pl) * 2 + (ci % 16) / 8) * d.Cout_pad + co) * 8 + ci % 8] = (uint16_t)(piece[pl] >> 16);
  };
  if (d.stride == 1) {
    // Conv1d weight [Cout][Cin][K]
    IX_ARG(shape[0] == d.Cout && shape[1] == d.Cin && shape[2] == d.K, "bigvgan_set_tensor: %s expects [%d,%d,%d]", name, d.Cout, d.Cin, d.K);
    for (int co = 0; co < d.Cout; ++co)
      for (int ci = 0; ci < d.Cin; ++ci)
        for (int k = 0; k < d.K; ++k) put(k, ci, co, data[((size_t)co * d.Cin + ci) * d.K + k]);
  } else {
    // ConvTranspose1d weight [Cin][Cout][K] -> tap slot = phase r * (K/stride) + j, k = r + stride*j
    IX_ARG(shape[0] == d.Cin && shape[1] == d.Cout && shape[2] == d.K, "bigvgan_set_tensor: %s expects [%d,%d,%d]", name, d.Cin, d.Cout, d.K);
    const int J = d.K / d.stride;
    for (int ci = 0; ci < d.Cin; ++ci)
      for (int co = 0; co < d.Cout; ++co)
        for (int k = 0; k < d.K; ++k) put((k % d.stride) * J + k / d.stride, ci, co, data[((size_t)ci * d.Cout + co) * d.K + k]);
  }
  IX_HIP(hipMemcpy(h->arena + d.w_off, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
  d.w_set = true;
  return IXTTS_OK;
}

// 12-tap Kaiser-sinc (filter.py:30-62 with cutoff .25, half-width .3): computed in double,
// rounded to fp32.  The Python binding overwrites these with torch's fp32 taps via the
// pseudo-tensor "filter" so both sides use bit-identical coefficients.
static void default_filter(float* f) {
  const double taps[12] = {0.0020289646927267313, 0.009389465674757957, -0.0255434587597847, -0.057657383382320404,
                           0.12857258319854736,   0.44320979714393616,  0.44320979714393616,  0.12857258319854736,
                           -0.057657383382320404, -0.0255434587597847,  0.009389465674757957, 0.0020289646927267313};
  for (int i = 0; i < 12; ++i) f[i] = (float)taps[i];
}

extern "C" int ixtts_bigvgan_finalize(ixtts_bigvgan* h) {
  IX_ARG(h, "bigvgan_finalize: null handle");
  for (auto& kv : h->convs) {
    if (!kv.second.w_set || (kv.second.has_bias && !kv.second.b_set)) {
      set_error("bigvgan_finalize: tensor(s) of '%s' were not supplied", kv.first.c_str());
      return IXTTS_ERR_STATE;
    }
  }
  for (auto& kv : h->acts) {
    if (!kv.second.a_set || !kv.second.b_set) {
      set_error("bigvgan_finalize: alpha/beta of '%s' were not supplied", kv.first.c_str());
      return IXTTS_ERR_STATE;
    }
  }
  float f[16] = {0};
  default_filter(f);
  IX_HIP(hipMemcpy(h->arena + h->filt_off, f, sizeof(f), hipMemcpyHostToDevice));
  IX_TRY(ensure_workspace(h, 1, h->cfg.max_frames));
  h->finalized = true;
  return IXTTS_OK;
}

extern "C" int ixtts_bigvgan_arena(ixtts_bigvgan* h, void** ptr, size_t* bytes) {
  IX_ARG(h && ptr && bytes, "bigvgan_arena: null argument");
  *ptr = h->arena;
  *bytes = h->arena_floats * sizeof(float);
  return IXTTS_OK;
}

extern "C" int ixtts_bigvgan_adopt_arena(ixtts_bigvgan* h) {
  IX_ARG(h, "bigvgan_adopt_arena: null handle");
  IX_TRY(ensure_workspace(h, 1, h->cfg.max_frames));
  h->finalized = true;
  return IXTTS_OK;
}

static int run_act(ixtts_bigvgan* h, const std::string& name, const float* x, float* y, int B, int T, hipStream_t st) {
  const ActDesc& a = h->acts.at(name);
  const float* f = h->arena + h->filt_off;
  return launch_aa_snake(x, y, f, f, h->arena + a.a_off, h->arena + a.b_off, B, a.C, T, h->cfg.fast_sin != 0, st);
}

// the activation written as the next conv's x planes (x3 mode)
static int run_act_planes(ixtts_bigvgan* h, const std::string& name, const float* x, void* xp, int B, int T, hipStream_t st) {
  const ActDesc& a = h->acts.at(name);
  const float* f = h->arena + h->filt_off;
  return launch_aa_snake_planes(x, xp, f, f, h->arena + a.a_off, h->arena + a.b_off, B, a.C, T, h->cfg.fast_sin != 0, st);
}

// x: [B][Cin][Tin] fp32, or (x3 mode) the planes of that tensor
static int run_conv(ixtts_bigvgan* h, const std::string& name, const float* x, float* y, const float* res,
                    const float* accum, int div3, int B, int Tin, hipStream_t st, const float* accum2 = nullptr) {
  const ConvDesc& d = h->convs.at(name);
  ConvParams p;
  memset(&p, 0, sizeof(p));
  p.x = x;
  p.wp = h->arena + d.w_off;
  p.bias = d.has_bias ? h->arena + d.b_off : nullptr;
  p.zeros = h->arena + h->zero_off;
  p.res = res;
  p.accum = accum;
  p.accum2 = accum2;
  p.y = y;
  p.B = B;
  p.Cin = d.Cin;
  p.Cin_pad = d.Cin_pad;
  p.Cout = d.Cout;
  p.Cout_pad = d.Cout_pad;
  p.Tin = Tin;
  p.div3 = div3;
  if (d.stride == 1) {
    p.Tout = Tin;
    p.ntap = d.K;
    p.dil = d.dil;
    p.off0 = -d.pad;
    p.os = 1;
    p.oo = 0;
    p.Nq = Tin;
    p.nphase = 1;
  } else {
    // phase r: t = q*s + r - pad, input index q - j (j = 0..K/s-1)  => dil = -1, off0 = 0
    p.Tout = Tin * d.stride;
    p.ntap = d.K / d.stride;
    p.dil = -1;
    p.off0 = 0;
    p.os = d.stride;
    p.oo = -d.pad;
    p.Nq = Tin + p.ntap;  // q up to Tin-1+ (ntap-1) still touches valid inputs; extra column is masked
    p.nphase = d.stride;
  }
  static const bool timing = getenv("IXTTS_BV_TIMING") != nullptr;  // developer table: per conv shape, time and TFLOP/s
  auto launch = [&]() { return h->x3 ? launch_conv1d_x3(p, st) : launch_conv1d(p, st); };
  if (!timing) return launch();
  static std::map<std::string, std::pair<double, double>> table;  // shape -> (us, flops)
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0, st);
  const int rc = launch();
  hipEventRecord(e1, st);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  char key[128];
  snprintf(key, sizeof(key), "Cin %4d Cout %4d k %2d dil %2d stride %d T %7d", d.Cin, d.Cout, d.K, d.dil, d.stride, p.Tout);
  auto& t = table[key];
  t.first += ms * 1e3;
  t.second += 2.0 * d.Cin * d.Cout * d.K * (double)Tin * B;
  if (name == "resblocks." + std::to_string(h->cfg.n_stages * h->cfg.n_resblock_kernels - 1) + ".convs2.2") {
    double tu = 0, tf = 0;
    for (auto& kv : table) {
      fprintf(stderr, "[bv] %s : %9.1f us  %6.1f TFLOP/s\n", kv.first.c_str(), kv.second.first, kv.second.second / kv.second.first / 1e6);
      tu += kv.second.first;
      tf += kv.second.second;
    }
    fprintf(stderr, "[bv] convs total %.1f us, %.1f TFLOP/s\n", tu, tf / tu / 1e6);
    table.clear();
  }
  return rc;
}

extern "C" int ixtts_bigvgan_forward(ixtts_bigvgan* h, const float* mel, int B, int F, float* wav, void* stream) {
  IX_ARG(h && (mel || F == 0) && (wav || F == 0), "bigvgan_forward: null argument");
  if (!h->finalized) { set_error("bigvgan_forward: handle not finalized"); return IXTTS_ERR_STATE; }
  IX_ARG(B >= 0 && F >= 0, "bigvgan_forward: negative shape");
  if (B == 0 || F == 0) return IXTTS_OK;
  hipStream_t st = (hipStream_t)stream;
  IX_TRY(ensure_workspace(h, B, F));
  float *XS = h->buf[0], *X = h->buf[1];
  float* T1 = h->buf[3];
  const ixtts_bigvgan_cfg& c = h->cfg;
  static const bool timing = getenv("IXTTS_BV_TIMING") != nullptr;
  const bool conc = h->concurrent && !timing;
  if (conc && !h->ev_fork) {
    for (int k = 0; k < 2; ++k) {
      IX_HIP(hipStreamCreateWithFlags(&h->side[k], hipStreamNonBlocking));
      IX_HIP(hipEventCreateWithFlags(&h->ev_join[k], hipEventDisableTiming));
    }
    IX_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
  }
  // conv input of a resblock conv: the Snake pass writes it -- fp32 rows for conv1d.hip, the x planes for conv1d_x3.hip
  auto act_to = [&](const std::string& name, const float* src, float* f32dst, void* planes, int T_, hipStream_t s_) {
    return h->x3 ? run_act_planes(h, name, src, planes, B, T_, s_) : run_act(h, name, src, f32dst, B, T_, s_);
  };
  // conv input that no Snake pass produces (the mel, a stage output): split pass in x3 mode
  auto conv_in = [&](const float* src, int C_, int T_) -> const float* {
    if (!h->x3) return src;
    launch_split_planes(src, h->pbuf[3], B, C_, T_, st);
    return reinterpret_cast<const float*>(h->pbuf[3]);
  };
  IX_TRY(run_conv(h, "conv_pre", conv_in(mel, c.num_mels, F), XS, nullptr, nullptr, 0, B, F, st));
  int T = F;
  int ch = c.upsample_initial_channel;
  for (int i = 0; i < c.n_stages; ++i) {
    IX_TRY(run_conv(h, "ups." + std::to_string(i) + ".0", conv_in(XS, ch, T), X, nullptr, nullptr, 0, B, T, st));
    T *= c.upsample_rates[i];
    ch /= 2;
    if (conc) IX_HIP(hipEventRecord(h->ev_fork, st));
    // branch j works in (R_j, T1_j, T2_j); the longest one (the last kernel size) stays on the caller's stream and forms the mean
    for (int j = 0; j < c.n_resblock_kernels; ++j) {
      const bool last = j == c.n_resblock_kernels - 1;
      hipStream_t bs = (conc && !last) ? h->side[j] : st;
      float *Rj = h->buf[2 + 3 * j], *T1j = h->buf[3 + 3 * j], *T2j = h->buf[4 + 3 * j];
      const float* Aj = h->x3 ? reinterpret_cast<const float*>(h->pbuf[j]) : T1j;  // what the convs of this branch read
      if (conc && !last) IX_HIP(hipStreamWaitEvent(bs, h->ev_fork, 0));
      std::string p = "resblocks." + std::to_string(i * c.n_resblock_kernels + j);
      const float* cur = X;
      for (int m = 0; m < 3; ++m) {
        IX_TRY(act_to(p + ".activations." + std::to_string(2 * m) + ".act", cur, T1j, h->pbuf[j], T, bs));
        IX_TRY(run_conv(h, p + ".convs1." + std::to_string(m), Aj, T2j, nullptr, nullptr, 0, B, T, bs));
        IX_TRY(act_to(p + ".activations." + std::to_string(2 * m + 1) + ".act", T2j, T1j, h->pbuf[j], T, bs));
        if (m < 2 || !last) {
          IX_TRY(run_conv(h, p + ".convs2." + std::to_string(m), Aj, Rj, cur, nullptr, 0, B, T, bs));
          cur = Rj;
        } else {
          // xs = r0 ; xs += r1 ; x = (xs + r2) / 3      (bigvgan.py:369-375)
          if (conc)
            for (int k = 0; k < 2; ++k) IX_HIP(hipStreamWaitEvent(st, h->ev_join[k], 0));
          IX_TRY(run_conv(h, p + ".convs2." + std::to_string(m), Aj, XS, cur, h->buf[2], 1, B, T, st, h->buf[5]));
        }
      }
      if (conc && !last) IX_HIP(hipEventRecord(h->ev_join[j], bs));
    }
  }
  IX_TRY(run_act(h, "activation_post.act", XS, T1, B, T, st));
  const ConvDesc& cp = h->convs.at("conv_post");
  IX_TRY(launch_conv_post(T1, h->arena + cp.w_off, nullptr, wav, B, cp.Cin, T, st));
  return IXTTS_OK;
}

extern "C" double ixtts_bigvgan_flops(const ixtts_bigvgan* h, int B, int F) {
  if (!h) return 0.0;
  const ixtts_bigvgan_cfg& c = h->cfg;
  double fl = 2.0 * c.num_mels * c.upsample_initial_channel * 7 * F;
  double T = F;
  int ch = c.upsample_initial_channel;
  for (int i = 0; i < c.n_stages; ++i) {
    fl += 2.0 * ch * (ch / 2) * c.upsample_kernel_sizes[i] * T;
    T *= c.upsample_rates[i];
    ch /= 2;
    for (int j = 0; j < c.n_resblock_kernels; ++j) fl += 6.0 * 2.0 * ch * ch * c.resblock_kernel_sizes[j] * T;
  }
  fl += 2.0 * ch * 7 * T;
  return fl * B;
}

extern "C" int ixtts_bigvgan_destroy(ixtts_bigvgan* h) {
  if (!h) return IXTTS_OK;
  if (h->arena) hipFree(h->arena);
  for (int i = 0; i < ixtts_bigvgan::NBUF; ++i)
    if (h->buf[i]) hipFree(h->buf[i]);
  for (int i = 0; i < 4; ++i)
    if (h->pbuf[i]) hipFree(h->pbuf[i]);
  for (int k = 0; k < 2; ++k) {
    if (h->side[k]) hipStreamDestroy(h->side[k]);
    if (h->ev_join[k]) hipEventDestroy(h->ev_join[k]);
  }
  if (h->ev_fork) hipEventDestroy(h->ev_fork);
  delete h;
  return IXTTS_OK;
}
