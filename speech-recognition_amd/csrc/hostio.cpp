// Host-side input decoding of libasr_mi355x (no GPU work): CRC-32C for TFRecord framing and the audio
// file decoders that feed the device front end.  The reference reads audio with tensorflow-io
// (data.py:94-117): 16-bit WAV / FLAC / raw PCM -> float32 / 32768, channels averaged.  These are
// native replacements so that the input pipeline does not run sample loops in Python.
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../include/asr_mi355x.h"

void asr_set_error(const char* fmt, ...);

#define IO_CHECK(cond, code, ...) \
  do {                            \
    if (!(cond)) {                \
      asr_set_error(__VA_ARGS__); \
      return (code);              \
    }                             \
  } while (0)

// ------------------------------------------------------------------------------------------ CRC-32C
static uint32_t g_crc_tab[8][256];
static bool g_crc_ready = false;

static void crc_init() {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1u) ? 0x82F63B78u : 0u);
    g_crc_tab[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i)
    for (int t = 1; t < 8; ++t) g_crc_tab[t][i] = (g_crc_tab[t - 1][i] >> 8) ^ g_crc_tab[0][g_crc_tab[t - 1][i] & 0xFF];
  g_crc_ready = true;
}

extern "C" uint32_t asr_crc32c(const void* data, long n, uint32_t crc) {
  if (!g_crc_ready) crc_init();   // idempotent table fill: a race writes the same values
  const uint8_t* p = static_cast<const uint8_t*>(data);
  uint32_t c = ~crc;
  while (n >= 8) {                // slicing-by-8
    uint32_t lo, hi;
    memcpy(&lo, p, 4);
    memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = g_crc_tab[7][lo & 0xFF] ^ g_crc_tab[6][(lo >> 8) & 0xFF] ^ g_crc_tab[5][(lo >> 16) & 0xFF] ^ g_crc_tab[4][lo >> 24] ^
        g_crc_tab[3][hi & 0xFF] ^ g_crc_tab[2][(hi >> 8) & 0xFF] ^ g_crc_tab[1][(hi >> 16) & 0xFF] ^ g_crc_tab[0][hi >> 24];
    p += 8;
    n -= 8;
  }
  while (n-- > 0) c = (c >> 8) ^ g_crc_tab[0][(c ^ *p++) & 0xFF];
  return ~c;
}

// ------------------------------------------------------------------------------------------ WAV / PCM
static inline uint32_t rd32le(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
static inline uint16_t rd16le(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

struct WavInfo { int channels, rate, bits; const uint8_t* data; long data_bytes; };

static int wav_parse(const uint8_t* f, long n, WavInfo* w) {
  IO_CHECK(n >= 12 && !memcmp(f, "RIFF", 4) && !memcmp(f + 8, "WAVE", 4), ASR_ERR_ARG, "wav: not a RIFF/WAVE file");
  long pos = 12;
  bool have_fmt = false;
  while (pos + 8 <= n) {
    const uint32_t size = rd32le(f + pos + 4);
    const uint8_t* body = f + pos + 8;
    if (!memcmp(f + pos, "fmt ", 4)) {
      IO_CHECK(size >= 16 && pos + 8 + 16 <= n, ASR_ERR_ARG, "wav: short fmt chunk");
      const int tag = rd16le(body);
      w->channels = rd16le(body + 2);
      w->rate = (int)rd32le(body + 4);
      w->bits = rd16le(body + 14);
      IO_CHECK(tag == 1 || tag == 0xFFFE, ASR_ERR_UNSUPPORTED, "wav: format tag %d is not PCM", tag);
      have_fmt = true;
    } else if (!memcmp(f + pos, "data", 4)) {
      IO_CHECK(have_fmt, ASR_ERR_ARG, "wav: data chunk before fmt chunk");
      w->data = body;
      w->data_bytes = (long)size <= n - pos - 8 ? (long)size : n - pos - 8;   // tolerate a short final chunk
      return ASR_OK;
    }
    pos += 8 + (long)size + (size & 1);
  }
  asr_set_error("wav: no data chunk");
  return ASR_ERR_ARG;
}

// ------------------------------------------------------------------------------------------ FLAC
namespace {
struct BitReader {
  const uint8_t* p;
  long n, pos = 0;      // byte position
  uint64_t acc = 0;     // bit accumulator (left-aligned consumption from the top of `bits`)
  int bits = 0;
  bool fail = false;
  BitReader(const uint8_t* d, long len) : p(d), n(len) {}
  inline void fill(int need) {
    while (bits < need) {
      uint64_t byte = 0;
      if (pos < n) byte = p[pos]; else fail = true;
      ++pos;
      acc = (acc << 8) | byte;
      bits += 8;
    }
  }
  inline uint32_t get(int k) {          // k in [0, 32]
    if (k == 0) return 0;
    fill(k);
    const uint32_t v = (uint32_t)((acc >> (bits - k)) & ((k == 32) ? 0xFFFFFFFFull : ((1ull << k) - 1)));
    bits -= k;
    return v;
  }
  inline int32_t get_signed(int k) {
    if (k == 0) return 0;
    const uint32_t v = get(k);
    const uint32_t sign = 1u << (k - 1);
    return (int32_t)((v ^ sign) - sign) ;
  }
  inline uint32_t unary() {             // number of 0 bits before the next 1 bit
    uint32_t q = 0;
    for (;;) {
      if (bits == 0) fill(8);
      if (fail) return q;
      const uint64_t window = acc & ((bits == 64) ? ~0ull : ((1ull << bits) - 1));
      if (window == 0) { q += bits; bits = 0; continue; }
      const int lead = __builtin_clzll(window) - (64 - bits);
      q += lead;
      bits -= lead + 1;
      return q;
    }
  }
  inline void align() { bits -= bits & 7; }
  inline long byte_pos() const { return pos - bits / 8; }
};

static uint8_t crc8_flac(const uint8_t* p, long n) {
  uint8_t c = 0;
  for (long i = 0; i < n; ++i) {
    c ^= p[i];
    for (int k = 0; k < 8; ++k) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : (c << 1));
  }
  return c;
}
static uint16_t crc16_flac(const uint8_t* p, long n) {
  uint16_t c = 0;
  for (long i = 0; i < n; ++i) {
    c ^= (uint16_t)(p[i] << 8);
    for (int k = 0; k < 8; ++k) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : (c << 1));
  }
  return c;
}

struct FlacInfo { int rate = 0, channels = 0, bits = 0; long total = 0; long audio_offset = 0; };

static int flac_header(const uint8_t* f, long n, FlacInfo* fi) {
  IO_CHECK(n >= 42 && !memcmp(f, "fLaC", 4), ASR_ERR_ARG, "flac: missing fLaC marker");
  long pos = 4;
  bool last = false, have_info = false;
  while (!last) {
    IO_CHECK(pos + 4 <= n, ASR_ERR_ARG, "flac: truncated metadata");
    last = (f[pos] & 0x80) != 0;
    const int type = f[pos] & 0x7F;
    const long len = ((long)f[pos + 1] << 16) | ((long)f[pos + 2] << 8) | f[pos + 3];
    pos += 4;
    IO_CHECK(pos + len <= n, ASR_ERR_ARG, "flac: truncated metadata block");
    if (type == 0) {
      IO_CHECK(len >= 34, ASR_ERR_ARG, "flac: short STREAMINFO");
      const uint8_t* s = f + pos;
      fi->rate = (s[10] << 12) | (s[11] << 4) | (s[12] >> 4);
      fi->channels = ((s[12] >> 1) & 7) + 1;
      fi->bits = (((s[12] & 1) << 4) | (s[13] >> 4)) + 1;
      fi->total = ((long)(s[13] & 0xF) << 32) | ((long)s[14] << 24) | ((long)s[15] << 16) | ((long)s[16] << 8) | s[17];
      have_info = true;
    }
    pos += len;
  }
  IO_CHECK(have_info, ASR_ERR_ARG, "flac: no STREAMINFO block");
  fi->audio_offset = pos;
  return ASR_OK;
}

static int flac_residual(BitReader& br, int blocksize, int order, int32_t* out) {
  const int method = (int)br.get(2);
  IO_CHECK(method < 2, ASR_ERR_UNSUPPORTED, "flac: reserved residual coding method");
  const int pbits = method == 0 ? 4 : 5, escape = method == 0 ? 15 : 31;
  const int porder = (int)br.get(4);
  const int parts = 1 << porder;
  IO_CHECK((blocksize >> porder) << porder == blocksize || porder == 0, ASR_ERR_ARG, "flac: partition order does not divide the block");
  int i = order;
  for (int part = 0; part < parts; ++part) {
    int count = (blocksize >> porder) - (part == 0 ? order : 0);
    IO_CHECK(count >= 0 && i + count <= blocksize, ASR_ERR_ARG, "flac: bad residual partition");
    const int param = (int)br.get(pbits);
    if (param == escape) {
      const int raw = (int)br.get(5);
      for (int k = 0; k < count; ++k) out[i++] = br.get_signed(raw);
    } else {
      for (int k = 0; k < count; ++k) {
        const uint32_t q = br.unary();
        const uint32_t v = (q << param) | br.get(param);
        out[i++] = (int32_t)(v >> 1) ^ -(int32_t)(v & 1);
      }
    }
    IO_CHECK(!br.fail, ASR_ERR_ARG, "flac: truncated residual");
  }
  return ASR_OK;
}

static int flac_subframe(BitReader& br, int blocksize, int bps, int32_t* out) {
  IO_CHECK(br.get(1) == 0, ASR_ERR_ARG, "flac: subframe padding bit set (lost sync)");
  const int type = (int)br.get(6);
  int wasted = 0;
  if (br.get(1)) wasted = (int)br.unary() + 1;
  bps -= wasted;
  IO_CHECK(bps > 0 && bps <= 32, ASR_ERR_ARG, "flac: bad sample size");
  if (type == 0) {
    const int32_t v = br.get_signed(bps);
    for (int i = 0; i < blocksize; ++i) out[i] = v;
  } else if (type == 1) {
    for (int i = 0; i < blocksize; ++i) out[i] = br.get_signed(bps);
  } else if (type >= 8 && type <= 12) {
    const int order = type - 8;
    IO_CHECK(order <= blocksize, ASR_ERR_ARG, "flac: fixed order > block size");
    for (int i = 0; i < order; ++i) out[i] = br.get_signed(bps);
    const int rc = flac_residual(br, blocksize, order, out);
    if (rc != ASR_OK) return rc;
    for (int i = order; i < blocksize; ++i) {
      int64_t pred = 0;
      switch (order) {
        case 1: pred = out[i - 1]; break;
        case 2: pred = 2 * (int64_t)out[i - 1] - out[i - 2]; break;
        case 3: pred = 3 * (int64_t)out[i - 1] - 3 * (int64_t)out[i - 2] + out[i - 3]; break;
        case 4: pred = 4 * (int64_t)out[i - 1] - 6 * (int64_t)out[i - 2] + 4 * (int64_t)out[i - 3] - out[i - 4]; break;
        default: break;
      }
      out[i] = (int32_t)(out[i] + pred);
    }
  } else if (type >= 32) {
    const int order = (type & 31) + 1;
    IO_CHECK(order <= blocksize, ASR_ERR_ARG, "flac: LPC order > block size");
    for (int i = 0; i < order; ++i) out[i] = br.get_signed(bps);
    const int precision = (int)br.get(4) + 1;
    IO_CHECK(precision != 16, ASR_ERR_ARG, "flac: invalid LPC precision");
    const int shift = br.get_signed(5);
    IO_CHECK(shift >= 0, ASR_ERR_UNSUPPORTED, "flac: negative LPC shift");
    int32_t coef[32];
    for (int j = 0; j < order; ++j) coef[j] = br.get_signed(precision);
    const int rc = flac_residual(br, blocksize, order, out);
    if (rc != ASR_OK) return rc;
    for (int i = order; i < blocksize; ++i) {
      int64_t pred = 0;
      for (int j = 0; j < order; ++j) pred += (int64_t)coef[j] * out[i - 1 - j];
      out[i] = (int32_t)(out[i] + (pred >> shift));
    }
  } else {
    asr_set_error("flac: reserved subframe type %d", type);
    return ASR_ERR_UNSUPPORTED;
  }
  if (wasted)
    for (int i = 0; i < blocksize; ++i) out[i] = (int32_t)((uint32_t)out[i] << wasted);
  IO_CHECK(!br.fail, ASR_ERR_ARG, "flac: truncated subframe");
  return ASR_OK;
}

// decodes every frame; calls sink(channel_samples[ch][i], blocksize) per frame
template <class Sink>
static int flac_decode(const uint8_t* f, long n, const FlacInfo& fi, Sink&& sink) {
  long pos = fi.audio_offset;
  std::vector<int32_t> buf;
  static const int kBlock[16] = {0, 192, 576, 1152, 2304, 4608, -8, -16, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768};
  static const int kBits[8] = {0, 8, 12, -1, 16, 20, 24, 32};
  while (pos + 2 <= n) {
    if (!(f[pos] == 0xFF && (f[pos + 1] & 0xFE) == 0xF8)) {   // trailing padding / ID3: stop at lost sync after >= 1 frame
      break;
    }
    BitReader br(f + pos, n - pos);
    br.get(16);
    const int bs_code = (int)br.get(4), sr_code = (int)br.get(4);
    const int ch_code = (int)br.get(4), sz_code = (int)br.get(3);
    IO_CHECK(br.get(1) == 0, ASR_ERR_ARG, "flac: reserved frame-header bit set");
    // UTF-8 style coded frame / sample number (value not needed)
    uint32_t first = br.get(8);
    int extra = 0;
    while (first & 0x80) { first <<= 1; ++extra; }
    for (int k = 1; k < extra; ++k) br.get(8);
    int blocksize = kBlock[bs_code];
    IO_CHECK(blocksize != 0, ASR_ERR_ARG, "flac: reserved block size code");
    if (blocksize == -8) blocksize = (int)br.get(8) + 1;
    else if (blocksize == -16) blocksize = (int)br.get(16) + 1;
    if (sr_code == 12) br.get(8);
    else if (sr_code == 13 || sr_code == 14) br.get(16);
    IO_CHECK(sr_code != 15, ASR_ERR_ARG, "flac: invalid sample rate code");
    int bps = kBits[sz_code] == 0 ? fi.bits : kBits[sz_code];
    IO_CHECK(bps > 0, ASR_ERR_ARG, "flac: reserved sample size code");
    const long header_len = br.byte_pos();
    const uint8_t crc8 = (uint8_t)br.get(8);
    IO_CHECK(!br.fail && crc8 == crc8_flac(f + pos, header_len), ASR_ERR_ARG, "flac: frame header CRC mismatch");
    int channels;
    if (ch_code < 8) channels = ch_code + 1;
    else { IO_CHECK(ch_code <= 10, ASR_ERR_ARG, "flac: reserved channel assignment"); channels = 2; }
    IO_CHECK(channels == fi.channels, ASR_ERR_UNSUPPORTED, "flac: channel count changes mid-stream");
    buf.resize((size_t)channels * blocksize);
    for (int c = 0; c < channels; ++c) {
      const bool side = (ch_code == 8 && c == 1) || (ch_code == 9 && c == 0) || (ch_code == 10 && c == 1);
      const int rc = flac_subframe(br, blocksize, bps + (side ? 1 : 0), buf.data() + (size_t)c * blocksize);
      if (rc != ASR_OK) return rc;
    }
    br.align();
    const long body_len = br.byte_pos();
    const uint16_t crc16 = (uint16_t)br.get(16);
    IO_CHECK(!br.fail && crc16 == crc16_flac(f + pos, body_len), ASR_ERR_ARG, "flac: frame CRC mismatch");
    int32_t* a = buf.data();
    int32_t* b = buf.data() + blocksize;
    if (ch_code == 8) for (int i = 0; i < blocksize; ++i) b[i] = a[i] - b[i];
    else if (ch_code == 9) for (int i = 0; i < blocksize; ++i) a[i] = a[i] + b[i];
    else if (ch_code == 10)
      for (int i = 0; i < blocksize; ++i) {
        const int32_t side = b[i];
        const int32_t mid = (int32_t)(((uint32_t)a[i] << 1) | (side & 1));
        a[i] = (mid + side) >> 1;
        b[i] = (mid - side) >> 1;
      }
    sink(buf.data(), channels, blocksize);
    pos += br.byte_pos();
  }
  return ASR_OK;
}
}  // namespace

// ------------------------------------------------------------------------------------------ C ABI
extern "C" int asr_audio_info(const uint8_t* file, long nbytes, int format, asr_audio_info_t* info) {
  IO_CHECK(file && info && nbytes >= 0, ASR_ERR_ARG, "asr_audio_info: null argument");
  memset(info, 0, sizeof(*info));
  if (format == ASR_AUDIO_PCM16) {
    info->channels = 1; info->bits_per_sample = 16; info->sample_rate = 0; info->frames = (nbytes + 1) / 2;
    return ASR_OK;
  }
  if (format == ASR_AUDIO_WAV) {
    WavInfo w{};
    const int rc = wav_parse(file, nbytes, &w);
    if (rc != ASR_OK) return rc;
    IO_CHECK(w.channels > 0 && w.bits > 0, ASR_ERR_ARG, "wav: bad fmt chunk");
    info->channels = w.channels; info->bits_per_sample = w.bits; info->sample_rate = w.rate;
    info->frames = w.data_bytes / ((long)w.channels * (w.bits / 8));
    return ASR_OK;
  }
  if (format == ASR_AUDIO_FLAC) {
    FlacInfo fi;
    const int rc = flac_header(file, nbytes, &fi);
    if (rc != ASR_OK) return rc;
    info->channels = fi.channels; info->bits_per_sample = fi.bits; info->sample_rate = fi.rate; info->frames = fi.total;
    if (fi.total == 0) {   // unknown length in STREAMINFO: count by decoding
      long count = 0;
      const int rc2 = flac_decode(file, nbytes, fi, [&](const int32_t*, int, int bs) { count += bs; });
      if (rc2 != ASR_OK) return rc2;
      info->frames = count;
    }
    return ASR_OK;
  }
  asr_set_error("asr_audio_info: unknown format %d", format);
  return ASR_ERR_UNSUPPORTED;
}

extern "C" int asr_audio_decode(const uint8_t* file, long nbytes, int format, float* out, long capacity, long* n_out) {
  IO_CHECK(file && out && n_out, ASR_ERR_ARG, "asr_audio_decode: null argument");
  *n_out = 0;
  const float scale = 1.0f / 32768.0f;
  if (format == ASR_AUDIO_PCM16) {      // data.py:100-105: raw little-endian int16, an odd trailing byte is zero-extended
    const long frames = (nbytes + 1) / 2;
    IO_CHECK(frames <= capacity, ASR_ERR_SHAPE, "asr_audio_decode: need %ld floats, have %ld", frames, capacity);
    for (long i = 0; i < frames; ++i) {
      const uint8_t lo = file[2 * i], hi = 2 * i + 1 < nbytes ? file[2 * i + 1] : 0;
      out[i] = (float)(int16_t)(lo | (hi << 8)) * scale;
    }
    *n_out = frames;
    return ASR_OK;
  }
  if (format == ASR_AUDIO_WAV) {
    WavInfo w{};
    const int rc = wav_parse(file, nbytes, &w);
    if (rc != ASR_OK) return rc;
    IO_CHECK(w.bits == 16, ASR_ERR_UNSUPPORTED, "wav: %d-bit samples (the reference reads int16 only, data.py:98)", w.bits);
    IO_CHECK(w.channels > 0, ASR_ERR_ARG, "wav: zero channels");
    const long frames = w.data_bytes / (2L * w.channels);
    IO_CHECK(frames <= capacity, ASR_ERR_SHAPE, "asr_audio_decode: need %ld floats, have %ld", frames, capacity);
    for (long i = 0; i < frames; ++i) {
      float acc = 0.f;                   // data.py:99,116: cast / 32768 per channel, then mean over channels
      for (int c = 0; c < w.channels; ++c) acc += (float)(int16_t)rd16le(w.data + 2 * (i * w.channels + c)) * scale;
      out[i] = w.channels == 1 ? acc : acc / (float)w.channels;
    }
    *n_out = frames;
    return ASR_OK;
  }
  if (format == ASR_AUDIO_FLAC) {
    FlacInfo fi;
    const int rc = flac_header(file, nbytes, &fi);
    if (rc != ASR_OK) return rc;
    IO_CHECK(fi.bits == 16, ASR_ERR_UNSUPPORTED, "flac: %d-bit samples (the reference reads int16 only, data.py:98)", fi.bits);
    long count = 0;
    bool overflow = false;
    const int rc2 = flac_decode(file, nbytes, fi, [&](const int32_t* s, int ch, int bs) {
      if (count + bs > capacity) { overflow = true; count += bs; return; }
      for (int i = 0; i < bs; ++i) {
        float acc = 0.f;
        for (int c = 0; c < ch; ++c) acc += (float)s[(size_t)c * bs + i] * scale;
        out[count + i] = ch == 1 ? acc : acc / (float)ch;
      }
      count += bs;
    });
    if (rc2 != ASR_OK) return rc2;
    IO_CHECK(!overflow, ASR_ERR_SHAPE, "asr_audio_decode: need %ld floats, have %ld", count, capacity);
    if (fi.total > 0 && count > fi.total) count = fi.total;   // the last block may be padded
    *n_out = count;
    return ASR_OK;
  }
  asr_set_error("asr_audio_decode: unknown format %d", format);
  return ASR_ERR_UNSUPPORTED;
}
