// Scene scores and cut rules behind the C ABI: what stands between the integer sums
// of K1 / K2 and the values the reference observes from its ffmpeg child (model_manager.py:736-786: `select='gt(scene,T)'`,
// the pts_time of the frames that pass) and from PySceneDetect's ContentDetector (north star).  A handful of float64
// operations per frame, in the libraries' own order [PUBLIC-LIB, SURVEY.md appendix A.1 / A.2]; plain host C++ (this
// file holds no kernel), so that a binder in any language gets the same bits as eioku_amd/scene.py - which calls these
// entry points - without restating libavfilter's double -> float32 clip or the cut filter.
#include <cmath>
#include <vector>

#include "common.h"

using namespace eioku;

extern "C" {

// libavfilter get_scene_score over a SAD series: mafd = sad / count / 2^(bitdepth - 8);
// score = clip(float32(min(mafd, |mafd - prev_mafd|) / 100), 0, 1).  first_has_prev = 0: frame 0 has no predecessor
// (mafd[0] = score[0] = 0 and frame 1 is scored against prev_mafd, the filter's zero initialisation).
int eioku_scene_scores_from_sad(const uint64_t* sad, int n, double count, int bitdepth, double prev_mafd, int first_has_prev,
                                double* mafd_out, double* score_out) {
  EIOKU_REQUIRE(n >= 0 && count > 0 && bitdepth >= 8 && bitdepth <= 16, "bad argument");
  if (n == 0) return EIOKU_OK;
  EIOKU_REQUIRE(sad && score_out, "NULL pointer");
  const double scale = (double)(1u << (bitdepth - 8));
  std::vector<double> mafd((size_t)n);
  for (int t = 0; t < n; ++t) mafd[t] = (double)sad[t] / count / scale;
  auto clipf = [](double q) {
    float f = (float)q;  // av_clipf takes a float: the quotient is rounded to float32 first
    if (f < 0.0f) f = 0.0f;
    if (f > 1.0f) f = 1.0f;
    return (double)f;
  };
  for (int t = 0; t < n; ++t) {
    const double prev = t == 0 ? prev_mafd : mafd[t - 1];
    score_out[t] = clipf(fmin(mafd[t], fabs(mafd[t] - prev)) / 100.0);
  }
  if (!first_has_prev) {
    mafd[0] = 0.0;
    score_out[0] = 0.0;
    if (n > 1) score_out[1] = clipf(fmin(mafd[1], fabs(mafd[1] - prev_mafd)) / 100.0);
  }
  if (mafd_out)
    for (int t = 0; t < n; ++t) mafd_out[t] = mafd[t];
  return EIOKU_OK;
}

// K1 + the scores: luma planes (host or device, as eioku_scene_sad_luma) -> score_out[n] / mafd_out[n] (HOST doubles).
// Synchronises `stream`.  prev: the plane preceding frame 0 (NULL: none), prev_mafd: the mafd of that plane's frame.
int eioku_scene_scores_luma(const uint8_t* y_frames, int n, int h, int w, size_t row_stride, size_t frame_stride,
                            const uint8_t* prev, double prev_mafd, double* mafd_out, double* score_out, int mem,
                            void* stream_) {
  EIOKU_REQUIRE_INIT();
  if (n == 0) return EIOKU_OK;
  EIOKU_REQUIRE(score_out, "NULL output");
  std::vector<uint64_t> sad((size_t)n);
  int rc;
  if (mem == EIOKU_MEM_DEVICE) {
    uint64_t* d = (uint64_t*)scratch(kSlotWork2, sizeof(uint64_t) * (size_t)n);
    if (!d) return EIOKU_ENOMEM;
    rc = eioku_scene_sad_luma(y_frames, n, h, w, row_stride, frame_stride, prev, d, mem, stream_);
    if (rc) return rc;
    EIOKU_HIP_CHECK(hipMemcpyAsync(sad.data(), d, sizeof(uint64_t) * (size_t)n, hipMemcpyDeviceToHost, (hipStream_t)stream_));
  } else {
    rc = eioku_scene_sad_luma(y_frames, n, h, w, row_stride, frame_stride, prev, sad.data(), mem, stream_);
    if (rc) return rc;
  }
  EIOKU_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_));
  return eioku_scene_scores_from_sad(sad.data(), n, (double)h * (double)w, 8, prev_mafd, prev != nullptr, mafd_out, score_out);
}

// ContentDetector frame score from K2's sums [n][3] (hue, sat, val): (dh * 1 + ds * 1 + dl * 1 + 0 * 0) / 3 with
// d = sum / float(pixels), accumulated in that order from 0.0 (PySceneDetect's weighted sum with the default weights).
int eioku_scene_content_scores(const uint64_t* sums, int n, double num_pixels, int first_has_prev, double* score_out) {
  EIOKU_REQUIRE(n >= 0 && num_pixels > 0, "bad argument");
  if (n == 0) return EIOKU_OK;
  EIOKU_REQUIRE(sums && score_out, "NULL pointer");
  for (int t = 0; t < n; ++t) {
    double acc = 0.0 + ((double)sums[3 * t] / num_pixels) * 1.0;
    acc = acc + ((double)sums[3 * t + 1] / num_pixels) * 1.0;
    acc = acc + ((double)sums[3 * t + 2] / num_pixels) * 1.0;
    acc = acc + 0.0 * 0.0;
    score_out[t] = acc / 3.0;
  }
  if (!first_has_prev) score_out[0] = 0.0;
  return EIOKU_OK;
}

// Cut frames from ContentDetector scores.  mode 0: PySceneDetect 0.6.0-0.6.3 (score >= threshold and min_scene_len
// frames since the last cut, counted from the first frame; the same rule is 0.6.4+'s FlashFilter SUPPRESS);
// mode 1: 0.6.4+ FlashFilter.Mode.MERGE.  Frame 0 never cuts.  cuts_out[cap]; *n_cuts = cuts found (may exceed cap:
// then only the first cap were written and the call returns EIOKU_EINVAL).
int eioku_scene_content_cuts(const double* scores, int n, double threshold, int min_scene_len, int mode, int32_t* cuts_out,
                             int cap, int* n_cuts) {
  EIOKU_REQUIRE(n >= 0 && cap >= 0 && n_cuts && (mode == 0 || mode == 1), "bad argument");
  EIOKU_REQUIRE(n == 0 || scores, "NULL scores");
  EIOKU_REQUIRE(cap == 0 || cuts_out, "NULL cuts_out");
  int found = 0;
  auto emit = [&](int t) {
    if (found < cap) cuts_out[found] = t;
    ++found;
  };
  auto above = [&](int t) { return t > 0 && scores[t] >= threshold; };
  if (mode == 0) {
    int last = 0;
    for (int t = 0; t < n; ++t)
      if (above(t) && t - last >= min_scene_len) {
        emit(t);
        last = t;
      }
  } else if (min_scene_len <= 0) {
    for (int t = 0; t < n; ++t)
      if (above(t)) emit(t);
  } else {
    int last_above = 0, start = 0;  // the filter first sees frame 0
    bool enabled = false, triggered = false;
    for (int t = 0; t < n; ++t) {
      const bool met = (t - last_above) >= min_scene_len;
      const bool ab = above(t);
      if (ab) last_above = t;
      if (triggered) {
        if (met && !ab && (last_above - start) >= min_scene_len) {
          triggered = false;
          emit(last_above);
        }
        continue;
      }
      if (!ab) continue;
      if (met) {
        enabled = true;
        emit(t);
      } else if (enabled) {
        triggered = true;
        start = t;
      }
    }
  }
  *n_cuts = found;
  EIOKU_REQUIRE(found <= cap, "%d cuts found, room for %d", found, cap);
  return EIOKU_OK;
}

// K2 + score + cut rule in one call (SURVEY.md 8b's eioku_scene_content): BGR frames (host or device) -> cut frame
// indices; score_out (optional, HOST, n doubles).  Synchronises `stream`.
int eioku_scene_content(const uint8_t* bgr_frames, int n, int h, int w, const uint8_t* prev, double threshold,
                        int min_scene_len, int mode, int32_t* cuts_out, int cap, int* n_cuts, double* score_out, int mem,
                        void* stream_) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(n_cuts, "NULL n_cuts");
  *n_cuts = 0;
  if (n == 0) return EIOKU_OK;
  std::vector<uint64_t> sums((size_t)n * 3);
  int rc;
  if (mem == EIOKU_MEM_DEVICE) {
    uint64_t* d = (uint64_t*)scratch(kSlotWork2, sizeof(uint64_t) * 3 * (size_t)n);
    if (!d) return EIOKU_ENOMEM;
    rc = eioku_scene_hsv_sums(bgr_frames, n, h, w, (size_t)h * w * 3, prev, d, mem, stream_);
    if (rc) return rc;
    EIOKU_HIP_CHECK(hipMemcpyAsync(sums.data(), d, sizeof(uint64_t) * 3 * (size_t)n, hipMemcpyDeviceToHost, (hipStream_t)stream_));
  } else {
    rc = eioku_scene_hsv_sums(bgr_frames, n, h, w, (size_t)h * w * 3, prev, sums.data(), mem, stream_);
    if (rc) return rc;
  }
  EIOKU_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_));
  std::vector<double> sc((size_t)n);
  rc = eioku_scene_content_scores(sums.data(), n, (double)h * (double)w, prev != nullptr, sc.data());
  if (rc) return rc;
  if (score_out)
    for (int t = 0; t < n; ++t) score_out[t] = sc[t];
  return eioku_scene_content_cuts(sc.data(), n, threshold, min_scene_len, mode, cuts_out, cap, n_cuts);
}

}  // extern "C"
