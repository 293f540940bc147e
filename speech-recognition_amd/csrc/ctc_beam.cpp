// CTC prefix beam search on the host: what tf.nn.ctc_beam_search_decoder does for
// DeepSpeechSearcher.beam_search (search.py:254-285).  TensorFlow runs this op on the CPU as well (there is no
// device kernel for it): it is a tree of label prefixes that grows by data-dependent pointer chasing, one frame
// after the other.  The device produces the log-probabilities (asr_ctc_log_softmax, search.hip); this file walks them.
//
// Algorithm ([TF-sem] CTCBeamSearchDecoder::Step / TopPaths, merge_repeated = false, default scorer, no label
// selection), per frame, all in float like TensorFlow's T = float instantiation:
//   every entry of the beam keeps log p_blank, p_label, p_total of its prefix;
//   1. entries in the beam are advanced in place:
//        p_label' = (p_label (+) prev) * P(label)   with prev = the parent's p_blank when the entry repeats the
//                                                   parent's label, else the parent's p_total (parent still active);
//        p_blank' = p_total * P(blank);  p_total' = p_blank' (+) p_label';
//   2. every entry whose old p_total can still compete grows its C-1 children that are not in the beam:
//        p_label = P(c) * (c == entry's label ? entry's old p_blank : entry's old p_total), p_blank = 0,
//      and a child enters the beam iff the beam is not full or its p_total beats the beam's worst (strictly),
//      in which case the worst entry leaves the beam (and becomes inactive).
// Unlike TensorFlow this does not materialise all C-1 children of an expanded entry (TensorFlow allocates them
// all; with a 16 k vocabulary that is gigabytes per utterance): a child record exists only once it has entered
// the beam.  The survivors are the same - a child that never enters carries no state - including TensorFlow's
// order-dependent corner: an entry pushed out of the beam during step 2 whose parent then re-examines it and
// finds it uncompetitive loses its old probabilities too, so it no longer grows children in that frame.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "../../include/asr_mi355x.h"

void asr_set_error(const char* fmt, ...);

#define ASR_HOST_CHECK(cond, code, ...) \
  do {                                  \
    if (!(cond)) {                      \
      asr_set_error(__VA_ARGS__);       \
      return (code);                    \
    }                                   \
  } while (0)

namespace {

constexpr float kLogZero = -std::numeric_limits<float>::infinity();

inline float log_sum_exp(float a, float b) {
  if (a == kLogZero) return b;
  if (b == kLogZero) return a;
  return a > b ? a + log1pf(expf(b - a)) : b + log1pf(expf(a - b));
}

struct Prob {
  float total = kLogZero, blank = kLogZero, label = kLogZero;
};

struct Entry {
  int parent;
  int label;
  Prob oldp, newp;
  std::vector<std::pair<int, int>> children;   // (label, entry index) of children that have been in the beam
  bool active() const { return newp.total != kLogZero; }
};

struct Decoder {
  int C, width;
  std::vector<Entry> pool;
  std::vector<int> leaves;                     // min-heap on newp.total
  std::vector<uint8_t> skip;

  Decoder(int classes, int beam_width) : C(classes), width(beam_width), skip((size_t)classes, 0) {
    pool.reserve(1024);
    Entry root;
    root.parent = -1;
    root.label = -1;
    root.newp.total = 0.f;
    root.newp.blank = 0.f;
    pool.push_back(root);
    leaves.push_back(0);
  }

  struct Worse {
    const std::vector<Entry>* pool;
    bool operator()(int a, int b) const { return (*pool)[a].newp.total > (*pool)[b].newp.total; }   // min-heap
  };

  void push_leaf(int e) {
    leaves.push_back(e);
    std::push_heap(leaves.begin(), leaves.end(), Worse{&pool});
  }
  int bottom() const { return leaves.front(); }
  void pop_bottom() {
    std::pop_heap(leaves.begin(), leaves.end(), Worse{&pool});
    leaves.pop_back();
  }
  bool is_candidate(float total) const {
    return total > kLogZero && ((int)leaves.size() < width || total > pool[bottom()].newp.total);
  }

  int child_of(int parent, int label) const {
    for (auto& ch : pool[parent].children)
      if (ch.first == label) return ch.second;
    return -1;
  }

  void step(const float* row) {
    const int blank = C - 1;
    float mx = row[0];
    for (int c = 1; c < C; ++c) mx = std::max(mx, row[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(row[c] - mx);
    const float norm = mx + logf(s);

    std::vector<int> branches = leaves;
    std::sort(branches.begin(), branches.end(), [&](int a, int b) {
      return pool[a].newp.total > pool[b].newp.total || (pool[a].newp.total == pool[b].newp.total && a < b);
    });
    leaves.clear();
    for (int bi : branches) pool[bi].oldp = pool[bi].newp;
    for (int bi : branches) {
      Entry& b = pool[bi];
      if (b.parent >= 0) {
        const Entry& par = pool[b.parent];
        if (par.active()) {
          const float prev = b.label == par.label ? par.oldp.blank : par.oldp.total;
          b.newp.label = log_sum_exp(b.newp.label, prev);
        }
        b.newp.label += row[b.label] - norm;
      }
      b.newp.blank = b.oldp.total + row[blank] - norm;
      b.newp.total = log_sum_exp(b.newp.blank, b.newp.label);
      push_leaf(bi);
    }
    for (int bi : branches) {
      if (!is_candidate(pool[bi].oldp.total)) continue;
      // skip: 1 = this child is in the beam, 2 = it has a record (it was in the beam once) but is inactive
      for (auto& ch : pool[bi].children) skip[ch.first] = pool[ch.second].active() ? 1 : 2;
      const float old_total = pool[bi].oldp.total, old_blank = pool[bi].oldp.blank;
      const int blabel = pool[bi].label;
      float floor_total = (int)leaves.size() < width ? kLogZero : pool[bottom()].newp.total;
      for (int c = 0; c < C - 1; ++c) {
        if (skip[c] == 1) continue;
        const float prev = c == blabel ? old_blank : old_total;
        const float lab = row[c] - norm + prev;
        if (!(lab > floor_total)) {                              // not a candidate (also lab == -inf)
          if (skip[c] == 2) pool[child_of(bi, c)].oldp = Prob();  // TensorFlow deactivates the child: oldp too
          continue;
        }
        if ((int)leaves.size() == width) {
          const int worst = bottom();
          pool[worst].newp = Prob();                             // the worst entry leaves the beam
          if (pool[worst].parent == bi) skip[pool[worst].label] = 2;
          pop_bottom();
        }
        int ci = skip[c] == 2 ? child_of(bi, c) : -1;
        if (ci < 0) {
          ci = (int)pool.size();
          Entry e;
          e.parent = bi;
          e.label = c;
          pool.push_back(e);
          pool[bi].children.emplace_back(c, ci);
        }
        Entry& ce = pool[ci];                                    // oldp is left as it is, like TensorFlow does
        ce.newp.blank = kLogZero;
        ce.newp.label = lab;
        ce.newp.total = lab;
        skip[c] = 1;
        push_leaf(ci);
        floor_total = (int)leaves.size() < width ? kLogZero : pool[bottom()].newp.total;
      }
      for (auto& ch : pool[bi].children) skip[ch.first] = 0;
    }
  }

  void top_paths(int n, int T, int32_t* tokens, int32_t* lengths, float* log_prob) {
    std::vector<int> best = leaves;
    std::sort(best.begin(), best.end(), [&](int a, int b) {
      return pool[a].newp.total > pool[b].newp.total || (pool[a].newp.total == pool[b].newp.total && a < b);
    });
    for (int i = 0; i < n; ++i) {
      int32_t* out = tokens + (long)i * T;
      std::memset(out, 0, sizeof(int32_t) * (size_t)T);
      if (i >= (int)best.size()) {
        lengths[i] = 0;
        log_prob[i] = kLogZero;
        continue;
      }
      int len = 0;
      for (int e = best[i]; pool[e].parent >= 0; e = pool[e].parent) ++len;
      int pos = len;
      for (int e = best[i]; pool[e].parent >= 0; e = pool[e].parent) out[--pos] = pool[e].label;
      lengths[i] = len;
      log_prob[i] = pool[best[i]].newp.total;
    }
  }
};

}  // namespace

extern "C" int asr_ctc_beam_search(const float* log_probs, int B, int T, int C, const int32_t* seq_len, int beam_width, int top_paths,
                                   int32_t* tokens, int32_t* lengths, float* log_prob, int threads) {
  ASR_HOST_CHECK(log_probs && tokens && lengths && log_prob, ASR_ERR_ARG, "asr_ctc_beam_search: null argument");
  ASR_HOST_CHECK(B > 0 && T > 0 && C >= 2 && beam_width > 0 && top_paths > 0 && top_paths <= beam_width, ASR_ERR_SHAPE,
                 "asr_ctc_beam_search: bad shape B=%d T=%d C=%d beam_width=%d top_paths=%d", B, T, C, beam_width, top_paths);
  if (seq_len)
    for (int b = 0; b < B; ++b)
      ASR_HOST_CHECK(seq_len[b] >= 0 && seq_len[b] <= T, ASR_ERR_SHAPE, "asr_ctc_beam_search: seq_len[%d]=%d outside [0, %d]", b, seq_len[b], T);
  auto work = [&](int b0, int b1) {
    for (int b = b0; b < b1; ++b) {
      Decoder d(C, beam_width);
      const int n = seq_len ? seq_len[b] : T;
      for (int t = 0; t < n; ++t) d.step(log_probs + ((long)b * T + t) * C);
      d.top_paths(top_paths, T, tokens + (long)b * top_paths * T, lengths + (long)b * top_paths, log_prob + (long)b * top_paths);
    }
  };
  const int nt = std::max(1, std::min(threads, B));
  if (nt == 1) {
    work(0, B);
  } else {
    std::vector<std::thread> pool;
    for (int i = 0; i < nt; ++i) pool.emplace_back(work, (int)((long)B * i / nt), (int)((long)B * (i + 1) / nt));
    for (auto& th : pool) th.join();
  }
  return ASR_OK;
}
