/*
 * eioku_hip.h - C ABI of libeioku_hip.so: the MI355X (gfx950) implementation of the
 * ml-service hot path of codihuston/eioku (scene scoring, YOLOv8 detection,
 * all-MiniLM-L6-v2 segment embedding, flat-L2 kNN).
 *
 * The reference (pure Python, /root/reference @ 2026-01-28) has no FFI for this path:
 * its arithmetic happens inside ffmpeg / OpenCV / Ultralytics / torchvision calls made
 * from ml-service/src/services/model_manager.py.  Each entry point below names the
 * reference call site whose work it replaces, so a maintainer can bind it (ctypes, see
 * INTEGRATION.md) behind the unchanged ModelManager.detect_* / process_ml_task API.
 *
 * Conventions
 *   - every function returns 0 on success or a negative EIOKU_E* code; the message for the
 *     calling thread's last failure is eioku_last_error().
 *   - `mem` says where the data pointers of THAT call live: EIOKU_MEM_HOST (the library
 *     stages through its own device buffers; PCIe-inclusive) or EIOKU_MEM_DEVICE (HBM
 *     pointers, zero copy).  Small result arrays follow the same flag.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls with
 *     EIOKU_MEM_DEVICE are asynchronous on that stream unless stated; EIOKU_MEM_HOST calls
 *     return after the results are in the host buffers.
 *   - handles are thread-compatible: one handle per thread, no internal locking.
 *   - no callbacks, no torch types, plain pointers and sizes only.
 */
#ifndef EIOKU_HIP_H
#define EIOKU_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EIOKU_ABI_VERSION 1

enum {
  EIOKU_OK = 0,
  EIOKU_EINVAL = -1,   /* bad argument (shape, alignment, NULL) */
  EIOKU_ENOMEM = -2,   /* device or host allocation failed */
  EIOKU_EHIP = -3,     /* a HIP runtime call failed; see eioku_last_error() */
  EIOKU_ENODEV = -4,   /* no gfx950 device / library not initialised */
  EIOKU_ESTATE = -5    /* handle used in the wrong state */
};

enum { EIOKU_MEM_HOST = 0, EIOKU_MEM_DEVICE = 1 };

/* ---- lifecycle ------------------------------------------------------------------------
 * Replaces: ModelManager._get_device()/gpu_available (model_manager.py:23-42): the Python
 * host keeps returning "cuda"; this selects the HIP device the worker process is pinned to.
 */
int eioku_abi_version(void);
int eioku_init(int device_id);
void eioku_shutdown(void);
const char* eioku_last_error(void);
/* name, CU count, HBM bytes of the active device (any pointer may be NULL). */
int eioku_device_info(char* name, size_t name_cap, int* compute_units, uint64_t* hbm_bytes);

/* ---- in-library kernel timing (bench.py roofline block) -------------------------------
 * When enabled, the launch sites of the tagged kernels are bracketed with hipEvents on the
 * stream they are launched on.  eioku_prof_read synchronises those events and returns the
 * summed duration and the number of launches since the last reset.
 */
enum {
  EIOKU_PROF_SCENE_SAD = 0,
  EIOKU_PROF_SCENE_HSV = 1,
  EIOKU_PROF_CONV = 2,
  EIOKU_PROF_KNN = 3,
  EIOKU_PROF_GEMM = 4,
  EIOKU_PROF_IVFPQ = 5, /* k_lscan, the list-major ADC scan */
  EIOKU_PROF_NUM_TAGS = 8
};
int eioku_prof_enable(int on); /* 0 off, 1 all tags, else mask: bit (tag + 1) enables that tag only */
int eioku_prof_reset(void);
int eioku_prof_read(int tag, double* total_ms, uint64_t* launches);

/* ---- synthetic inputs (bench / tests) ------------------------------------
 * Counter-based splitmix64, identical to oracle/prng.py.  Device pointers only.
 */
int eioku_synth_u64(uint64_t seed, uint64_t offset, uint64_t n, uint64_t* out_dev, void* stream);
int eioku_synth_bytes(uint64_t seed, uint64_t n, uint8_t* out_dev, void* stream);
/* approx-normal float32 (Irwin-Hall 4), optionally L2-normalised per row of `dim`. */
int eioku_synth_normal_f32(uint64_t seed, uint64_t rows, int dim, int l2_normalise,
                           float* out_dev, void* stream);
/* n BGR frames h x w; params_dev = n x 5 int32 {base_b,base_g,base_r,gx,gy}. */
int eioku_synth_frames_bgr(uint64_t seed, uint64_t first_frame, int n, int h, int w,
                           const int32_t* params_dev, uint8_t* out_dev, void* stream);

/* ---- scene scoring ---------------------------------------------------------------------
 * K1. Replaces the `ffmpeg -vf select='gt(scene\,T)',showinfo` child process of
 * ModelManager.detect_scenes (model_manager.py:736-755): per-frame luma SAD against the
 * previous frame, exact uint64.  y_frames: n planes of h rows, `row_stride` bytes apart,
 * planes `frame_stride` bytes apart.  prev (nullable): the plane preceding frame 0 (same
 * row_stride); without it sad_out[0] = 0.  sad_out: n uint64.
 */
int eioku_scene_sad_luma(const uint8_t* y_frames, int n, int h, int w, size_t row_stride,
                         size_t frame_stride, const uint8_t* prev, uint64_t* sad_out,
                         int mem, void* stream);

/* K2. PySceneDetect ContentDetector frame deltas (BASELINE.json north_star; the reference
 * holds intent only, .kiro/specs/semantic-video-search/design.md:59-61): OpenCV 8-bit
 * BGR->HSV, then per-frame sums of |c_t - c_{t-1}| for c = hue, sat, val, exact uint64.
 * bgr_frames: n packed frames (h*w*3 bytes each, `frame_stride` bytes apart).
 * sums_out: n x 3 uint64 (row 0 = 0 unless prev given).
 */
int eioku_scene_hsv_sums(const uint8_t* bgr_frames, int n, int h, int w, size_t frame_stride,
                         const uint8_t* prev, uint64_t* sums_out, int mem, void* stream);

/* K1 on decoded BGR frames, for the single-pass ingest entry (one upload of a frame serves scene + objects + faces):
 * luma = OpenCV's 8-bit COLOR_BGR2YUV_I420 luma of each pixel (BT.601 studio range, 20-bit fixed point - what
 * cv2.cvtColor gives for the frames cv2.VideoCapture.read() returns at model_manager.py:263), then the per-frame SAD of
 * eioku_scene_sad_luma.  Exact uint64.  h * w must be a multiple of 4.  The DECODER's own Y plane differs from this by
 * the YUV -> BGR -> Y round trip (+-1-2 codes on some pixels): use eioku_scene_sad_luma when that plane is available. */
int eioku_scene_sad_luma_bgr(const uint8_t* bgr_frames, int n, int h, int w, size_t frame_stride, const uint8_t* prev,
                             uint64_t* sad_out, int mem, void* stream);

/* ---- scene scores and cut rules (host arithmetic behind the ABI; csrc/scene_host.hip) ----
 * What the reference reads from its ffmpeg child (`select='gt(scene,T)'`: /root/reference/ml-service/src/services/
 * model_manager.py:736-786) and the ContentDetector the north star names, restated once, in the libraries' own float
 * order, so that a binder in any language gets the bits eioku_amd/scene.py gets.  Outputs are HOST doubles / ints.
 *   eioku_scene_scores_from_sad : libavfilter get_scene_score over a SAD series (mafd, clip(float32(min(mafd,
 *                                 |mafd - prev|) / 100))); count = pixels per plane.
 *   eioku_scene_scores_luma     : K1 + the above on luma planes (host or device); synchronises the stream.
 *   eioku_scene_content_scores  : ContentDetector score (dh + ds + dl + 0) / 3 from K2's sums [n][3].
 *   eioku_scene_content_cuts    : PySceneDetect cut filter; mode 0 = 0.6.0-0.6.3 / SUPPRESS, 1 = 0.6.4+ MERGE.
 *   eioku_scene_content         : K2 + score + cuts on BGR frames (host or device); synchronises the stream. */
int eioku_scene_scores_from_sad(const uint64_t* sad, int n, double count, int bitdepth, double prev_mafd, int first_has_prev,
                                double* mafd_out, double* score_out);
int eioku_scene_scores_luma(const uint8_t* y_frames, int n, int h, int w, size_t row_stride, size_t frame_stride,
                            const uint8_t* prev, double prev_mafd, double* mafd_out, double* score_out, int mem,
                            void* stream);
int eioku_scene_content_scores(const uint64_t* sums, int n, double num_pixels, int first_has_prev, double* score_out);
int eioku_scene_content_cuts(const double* scores, int n, double threshold, int min_scene_len, int mode, int32_t* cuts_out,
                             int cap, int* n_cuts);
int eioku_scene_content(const uint8_t* bgr_frames, int n, int h, int w, const uint8_t* prev, double threshold,
                        int min_scene_len, int mode, int32_t* cuts_out, int cap, int* n_cuts, double* score_out, int mem,
                        void* stream);

/* ---- decoder planes in, BGR on the device (csrc/yuv.hip) ----
 * OpenCV's 8-bit COLOR_YUV2BGR_I420 / _NV12 (BT.601 studio range, 20-bit fixed point): the conversion cap.read() runs on
 * the CPU inside the reference's frame loops (model_manager.py:237-297,331-398).  yuv: n frames of (3h/2) x w bytes in
 * OpenCV's planar Mat layout (layout 0 = I420: Y | U | V, 1 = NV12: Y | interleaved UV); bgr_out [n][h][w][3]; both on
 * the side `mem` names.  h even, w a multiple of 4.  The Y plane itself (rows 0 .. h-1 of every frame) is what
 * eioku_scene_sad_luma scores with frame_stride = 3 h w / 2. */
int eioku_yuv420_to_bgr(const uint8_t* yuv, int n, int h, int w, int layout, uint8_t* bgr_out, int mem, void* stream);

/* Debug / parity helper: the HSV image itself (same layout as the input). */
int eioku_bgr2hsv(const uint8_t* bgr, size_t n_pixels, uint8_t* hsv_out, int mem, void* stream);

/* ---- detection: YOLOv8 -----------------------------------------------------------------
 * K4 building block.  One NHWC fp16 convolution (k in {1,3}, stride in {1,2} (2 only for k=3),
 * pad k/2) as implicit GEMM on MFMA with fused bias + optional SiLU + optional residual.  This is
 * the arithmetic that `model(frame, ...)` (ultralytics, called at model_manager.py:270-275 and
 * :364-369) spends its time in (Conv2d + folded BatchNorm + SiLU).  Device pointers only.
 *   in_nhwc   : fp16, n x h x w pixels, `in_cstride` channels per pixel; the conv reads the `cin`
 *               channels starting at `in_coff` (a concat/chunk slice).  cin, strides, offsets % 8 == 0.
 *   weight    : HOST fp32 [cout][cin][k][k] (torch layout), bias HOST fp32 [cout] or NULL; rounded
 *               to fp16 like tensor.half().
 *   act_silu  : 0 none, 1 SiLU, 2 ReLU, 3 ReLU AFTER the residual sum (relu(fp16(conv + bias) + res): ResNet blocks).
 *   residual  : optional fp16 slice added after activation: out = fp16(fp16(act(..)) + res).
 *   out_nhwc  : fp16 slice (cstride/coff % 4 == 0), or out_f32 != NULL: dense fp32 [n,ho,wo,cout].
 * Synchronous (packs weights per call): a test / building-block entry, not the fast path.
 */
int eioku_conv2d_f16(const void* in_nhwc, int n, int h, int w, int in_cstride, int in_coff, int cin,
                     const float* weight_oihw, const float* bias, int cout, int ksize, int stride,
                     int act_silu, const void* residual, int res_cstride, int res_coff,
                     void* out_nhwc, int out_cstride, int out_coff, float* out_f32, void* stream);

/* Bounds-check instrumentation of the conv family (libeioku_hip_bc.so = the same sources with -DEIOKU_BOUNDS_CHECK: every
 * global access of an activation / residual / image / output tensor is compared with the tensor's extent, a violation is
 * counted, not performed).  violations = -1 in the regular library.  selftest != 0: first make 3 violations on purpose. */
int eioku_debug_bounds(int* violations, int* line, int reset, int selftest);

/* YOLOv8 detector handle.  Replaces `YOLO(model_path); model.to(device)` + `model(frame, conf=..)`
 * of ModelManager.detect_objects / detect_faces (model_manager.py:252-254,270-275 / :346-348,
 * :364-369).  The graph is the Ultralytics yolov8 layout parameterised by its backbone widths
 * ch5 = {c1..c5} and C2f repeats depth4 (n: {16,32,64,128,256},{1,2,2,1}; s: {32,..,512},{1,2,2,1};
 * m: {48,96,192,384,576},{2,4,4,2}); nc = 80 for COCO models, 1 for yolov8n-face.
 * Weights are set per convolution (BatchNorm already folded: conv + bias), indexed in module order;
 * eioku_yolo_conv_info returns the Ultralytics state-dict prefix ("model.2.m.0.cv1.conv", ...;
 * the three Detect output convs are "model.22.cv2.<l>.2" / "model.22.cv3.<l>.2") and the shape.
 * Convolution 0 takes 8 input channels: RGB in channels 0..2, channels 3..7 must be zero weights.
 */
typedef struct eioku_yolo eioku_yolo_t;
int eioku_yolo_create(const int* ch5, const int* depth4, int nc, eioku_yolo_t** out);
void eioku_yolo_destroy(eioku_yolo_t* y);
int eioku_yolo_num_convs(const eioku_yolo_t* y);
int eioku_yolo_conv_info(const eioku_yolo_t* y, int idx, char* name, size_t name_cap, int* cout, int* cin,
                         int* ksize, int* stride);
int eioku_yolo_set_conv(eioku_yolo_t* y, int idx, const float* weight_oihw, const float* bias);

/* Raw network: fp16 NHWC8 input (device, n x h x w x 8, h,w % 32 == 0) -> the six Detect maps
 * (device fp32): box_out[l] [n,h/s,w/s,64], cls_out[l] [n,h/s,w/s,nc], s = 8,16,32.  NULL entries
 * are skipped.  Asynchronous on `stream`. */
int eioku_yolo_forward(eioku_yolo_t* y, const void* in_nhwc8_f16, int n, int h, int w,
                       float* const* box_out, float* const* cls_out, void* stream);
/* Algorithmic conv FLOPs (2*Cout*Cin*k*k per output pixel, unpadded) of the last forward. */
int eioku_yolo_last_conv_flops(const eioku_yolo_t* y, double* flops);

/* One detection, ORIGINAL-frame pixels (after scale_boxes + clip), 32 bytes. */
typedef struct {
  float x1, y1, x2, y2;
  float conf;
  int32_t cls;
  int32_t anchor; /* index into the concatenated P3|P4|P5 anchor list: the "post-NMS box index" */
  int32_t pad;
} eioku_det_t;

/* Full per-frame pipeline on n BGR frames (u8, n x h x w x 3): K3 letterbox (cv2.resize
 * INTER_LINEAR fixed point / 2x2 area / copy, 114 padding, RGB, /255) -> network -> K6 decode ->
 * K7 class-aware NMS (IoU > iou suppresses, <= max_det kept, score order) -> scale_boxes + clip.
 * The letterbox geometry and the resize coefficient tables are produced by the host in the
 * reference's Python float semantics (eioku_amd/detect.py):
 *   lb_geom[9] = {new_h, new_w, top, left, out_h, out_w, mode(0 copy,1 bilinear,2 area), pad_x, pad_y}
 *   xofs[new_w], yofs[new_h] : first source column / row;  xalpha[new_w][2], ybeta[new_h][2] :
 *   11-bit fixed point tap weights (HOST pointers; may be NULL unless mode == 1).
 * dets_out: n x max_det eioku_det_t, counts_out: n int32 (host or device per `mem`; frames too).
 */
int eioku_yolo_detect(eioku_yolo_t* y, const uint8_t* bgr, int n, int h, int w, const int32_t* lb_geom,
                      const int32_t* xofs, const int32_t* yofs, const int16_t* xalpha,
                      const int16_t* ybeta, float gain, float conf, float iou, int max_det,
                      void* dets_out, int32_t* counts_out, int mem, void* stream);

/* Stage entry points (device pointers, asynchronous): K3 alone and K6+K7 alone.
 * eioku_letterbox_f16: BGR u8 frames -> fp16 NHWC8 (n x out_h x out_w x 8) network input.
 * eioku_yolo_postprocess: the six Detect maps -> detections, same semantics as eioku_yolo_detect
 * (hl/wl: map sizes of P3,P4,P5; gain/pad/src: scale_boxes parameters). */
int eioku_letterbox_f16(const uint8_t* bgr_dev, int n, int h, int w, const int32_t* lb_geom,
                        const int32_t* xofs, const int32_t* yofs, const int16_t* xalpha,
                        const int16_t* ybeta, void* out_nhwc8_dev, void* stream);
int eioku_yolo_postprocess(const float* const* box_dev, const float* const* cls_dev, int n, const int* hl,
                           const int* wl, int nc, float conf, float iou, int max_det, float gain,
                           int pad_x, int pad_y, int src_w, int src_h, void* dets_dev,
                           int32_t* counts_dev, void* stream);

/* ---- semantic search: exact kNN (FAISS IndexFlatL2 semantics) -----------------------------
 * The reference holds intent only for this stage (.kiro/specs/semantic-video-search/design.md:
 * 35-40,1105-1113; tasks.md:304-313 unchecked); BASELINE.json's north_star names FAISS IndexFlatL2.
 * search() returns SQUARED L2 distances ascending and int64 ids (-1 / FLT_MAX when fewer than k
 * vectors exist), ties broken by the smaller id.  d in {64,128,256,384,512}, k <= 32.
 * HBM layout: the fp32 rows [N][d] (+ fp32 norms) and, built lazily by the first search with nq > 64 over
 * >= scan_min_rows rows, a bf16 copy of the rows in MFMA-fragment order (2 d bytes per row).
 */
typedef struct eioku_index eioku_index_t;
int eioku_index_flat_create(int d, eioku_index_t** out);
void eioku_index_destroy(eioku_index_t* ix);
long long eioku_index_ntotal(const eioku_index_t* ix);
int eioku_index_reset(eioku_index_t* ix);
/* append n vectors (host: staged; device: copied into the index' own HBM buffer) */
int eioku_index_add(eioku_index_t* ix, const float* x, long long n, int mem, void* stream);
/* zero-copy: search an existing 16-byte aligned device buffer of n x d floats (bench: 10M x 384) */
int eioku_index_attach(eioku_index_t* ix, float* x_dev, long long n, void* stream);
int eioku_index_search(eioku_index_t* ix, const float* q, int nq, int k, float* D, int64_t* I, int mem,
                       void* stream);
/* the next k results AFTER a previous answer: only rows with (distance, id) > (after_D[q], after_I[q]) in the
 * result order are candidates (FAISS has no such call; IndexIVFPQ's probe selection uses it to lift nprobe above
 * the k <= 32 of one search: successive rounds on the exact-fp32 kernels see bit-identical distances). */
int eioku_index_search_after(eioku_index_t* ix, const float* q, int nq, int k, const float* after_D,
                             const int64_t* after_I, float* D, int64_t* I, int mem, void* stream);
/* Tuning / test knobs of the wide-search ("scan") path, see csrc/knn.hip: "scan_mode" 0 = register-tile kernels only,
 * 1 (default) = searches of >= "scan_min_nq" (default 1) queries over >= "scan_min_rows" rows keep row tiles stationary, filter with one bf16
 * product term and re-rank the candidates in fp32; "scan_cap" candidate slots per query (a list that overflows falls
 * back to the register-tile kernels); "scan_sample" rows of the bounding sample (0 = automatic); "scan_prescan" stride of the row tiles the
 * scan visits FIRST to tighten that bound (default 32; 0 = off, the sample alone bounds the scan); "scan_rt" 1 or 2
 * row tiles per wave (12 / 8 waves per workgroup). */
int eioku_index_set_param(eioku_index_t* ix, const char* name, long long value);
/* C1 helper: merge nlists per-shard results [nlists][nq][k] (ids already global) -> [nq][k].
 * Device pointers.  Used after the RCCL all-gather of per-rank (D, I). */
int eioku_topk_merge(const float* d_lists, const int64_t* i_lists, int nlists, int nq, int k, float* D,
                     int64_t* I, void* stream);

/* ---- C1 without a torch process group: sharded exact search over RCCL ----------------------------------
 * The reference holds intent only (.kiro/specs/semantic-video-search/design.md:58-61: one FAISS index per node);
 * BASELINE.json's north_star asks for "per-GPU FAISS-style shards ... merged via a single RCCL all-gather" behind
 * the C ABI.  Rank 0 makes an id (eioku_comm_unique_id) and ships the 128 bytes to its peers out of band (the
 * service's own RPC, a file, MPI ...); every rank, one process per GPU, then calls eioku_comm_create -- a
 * collective, like ncclCommInitRank, which it wraps -- and from then on eioku_index_search_sharded with the same
 * nq and k: local shard search, ONE all-gather of nq*k*12 bytes per rank over xGMI, local merge.  Every rank
 * receives the same global answer.  q / D / I are DEVICE pointers; ids are id_base + the shard's local row.
 * RCCL is bound with dlopen at first use (EIOKU_ENODEV if librccl.so is absent). */
#define EIOKU_COMM_ID_BYTES 128
typedef struct eioku_comm eioku_comm_t;
int eioku_comm_unique_id(unsigned char* id128);
int eioku_comm_create(const unsigned char* id128, int rank, int world, eioku_comm_t** out);
void eioku_comm_destroy(eioku_comm_t* comm);
int eioku_comm_rank(const eioku_comm_t* comm, int* rank, int* world);
int eioku_index_search_sharded(eioku_index_t* ix, eioku_comm_t* comm, long long id_base, const float* q, int nq,
                               int k, float* D, int64_t* I, void* stream);

/* ---- segment embedding: all-MiniLM-L6-v2 (BERT encoder + mean pooling + L2 norm) ----------
 * The reference holds intent only (.kiro/specs/semantic-video-search/design.md:54-57,1096-1103;
 * tasks.md:297-302 unchecked); BASELINE.json's north_star names all-MiniLM-L6-v2: vocab 30522,
 * hidden 384, 6 layers, 12 heads, ffn 1536, 512 positions, 2 token types, LayerNorm eps 1e-12.
 * Tensors carry the Hugging Face BertModel state-dict names ("embeddings.word_embeddings.weight",
 * "encoder.layer.0.attention.self.query.weight", ...), fp32, row-major [out][in].
 * Tokenisation (WordPiece) stays on the host.  head_dim must be 32; hidden, ffn % 128 == 0.
 */
typedef struct eioku_bert eioku_bert_t;
int eioku_bert_create(int vocab, int hidden, int layers, int heads, int ffn, int max_pos, int type_vocab,
                      float ln_eps, eioku_bert_t** out);
void eioku_bert_destroy(eioku_bert_t* m);
int eioku_bert_num_tensors(const eioku_bert_t* m);
int eioku_bert_tensor_info(const eioku_bert_t* m, int idx, char* name, size_t name_cap, int* rows, int* cols);
int eioku_bert_set_tensor(eioku_bert_t* m, int idx, const float* host_data, size_t numel);
/* ids int32 [B][S], mask uint8 [B][S] (1 = token) -> out float32 [B][hidden], unit L2 norm. */
int eioku_bert_embed(eioku_bert_t* m, const int32_t* ids, const uint8_t* mask, int B, int S, float* out,
                     int mem, void* stream);
int eioku_bert_last_flops(const eioku_bert_t* m, double* flops);

/* as eioku_topk_merge, for lists that carry k_in >= k entries each ([nlists][nq][k_in]) */
int eioku_topk_merge_ex(const float* d_lists, const int64_t* i_lists, int nlists, int nq, int k_in, int k,
                        float* D, int64_t* I, void* stream);

/* ---- approximate kNN: IVF-PQ building blocks (FAISS IndexIVFPQ semantics: L2, by_residual, 8 bit) ----
 * BASELINE.json cfg5.  Host orchestration (k-means / PQ training loops, list bookkeeping) lives in
 * eioku_amd/ivfpq.py; coarse assignment and probe selection reuse eioku_index_search (k = 1 / nprobe).
 * All pointers are DEVICE pointers; calls are asynchronous on `stream`.
 *   eioku_kmeans_update : centroids[k][d] <- mean of assigned rows (64-bit fixed-point atomics: bit
 *                         reproducible); empty clusters keep their centroid; counts_out optional.
 *   eioku_pq_assign     : nearest of 256 sub-centroids per (vector, sub-quantiser) of x - coarse[list]
 *                         (coarse may be NULL); codes_out [n][m] u8 and/or resid_out [n][d], d/m in {4,8,16}.
 *   eioku_ivf_histogram / eioku_ivf_scatter : counting sort of encoded vectors into inverted lists.
 *   eioku_ivfpq_scan    : ADC over the probed lists; partial results [nprobe][nq][K], K = 16 (k <= 16)
 *                         or 32, to be merged with eioku_topk_merge_ex.
 */
int eioku_kmeans_update(const float* x_dev, long long n, int d, const long long* assign_dev, int k,
                        float* centroids_dev, int* counts_out_dev, void* stream);
/* the two halves of eioku_kmeans_update for a build sharded over GPUs: every rank adds its rows
 * into caller-zeroed integer sums (int64 [k][d], 2^-32 fixed point) and counts (int32 [k]), the host all-reduces them
 * (RCCL; integers: exact, order independent) and every rank finalises the same centroids. */
int eioku_kmeans_accumulate(const float* x_dev, long long n, int d, const long long* assign_dev, int k,
                            long long* sums_dev, int* counts_dev, void* stream);
int eioku_kmeans_finalize(const long long* sums_dev, const int* counts_dev, int k, int d, float* centroids_dev,
                          void* stream);
int eioku_pq_assign(const float* x_dev, long long n, int d, int m, const float* coarse_dev,
                    const long long* list_dev, const float* pq_dev, uint8_t* codes_out_dev,
                    float* resid_out_dev, void* stream);
int eioku_ivf_histogram(const long long* list_dev, long long n, int nlist, int* counts_dev, void* stream);
int eioku_ivf_scatter(const long long* list_dev, long long n, int m, const uint8_t* codes_dev, long long id_base,
                      int* cursor_dev, uint8_t* list_codes_dev, long long* list_ids_dev, void* stream);
int eioku_ivfpq_scan(const float* q_dev, int nq, int d, int m, const long long* probes_dev, int nprobe,
                     const float* coarse_dev, const float* pq_dev, const int* offsets_dev,
                     const int* sizes_dev, const uint8_t* list_codes_dev, const long long* list_ids_dev,
                     int k, float* pd_dev, long long* pi_dev, void* stream);
/* FAISS' precomputed-table form of the same scan (IndexIVFPQ::use_precomputed_table): eioku_ivfpq_tables writes
 * out[v][j][c] = alpha ||pq[j][c]||^2 + beta (vecs[v]_j . pq[j][c]) as [nvec][m][256] floats - once per index over the
 * coarse centroids (alpha 1, beta 2: nlist x m x 1 KB) and once per search over the queries (alpha 0, beta -2) - and
 * eioku_ivfpq_scan_tables assembles each (query, list) look-up table as the sum of two rows + ||q - c||^2 instead of
 * m x 256 x dsub multiply-adds over the codebook.  Same results up to fp32 rounding of the decomposition. */
int eioku_ivfpq_tables(const float* vecs_dev, int nvec, int d, int m, const float* pq_dev, float alpha, float beta,
                       float* out_dev, void* stream);
int eioku_ivfpq_scan_tables(const float* q_dev, int nq, int d, int m, const long long* probes_dev, int nprobe,
                            const float* coarse_dev, const float* pq_dev, const int* offsets_dev,
                            const int* sizes_dev, const uint8_t* list_codes_dev, const long long* list_ids_dev,
                            const float* list_tables_dev, const float* query_tables_dev, int k, float* pd_dev,
                            long long* pi_dev, void* stream);

/* List-major form of the scan (round 3; d/m = 8, d in {64, 128, 256, 384}): the (query -> probes) relation is inverted per
 * search, a workgroup decodes a segment of ONE list once into bf16 MFMA operands and multiplies it with every query that
 * probes the list; the products only FILTER (rigorous bf16 margin against the k-th exact distance of the query's
 * nearest list), survivors are re-ranked with eioku_ivfpq_scan_tables' own fp32 arithmetic: (D, I) are bit-identical
 * to eioku_ivfpq_scan_tables + eioku_topk_merge_ex, codes cross HBM once per search instead of once per (query, probe).
 *   eioku_ivfpq_lists_aux       : index-side tables, once per pack: pqh_out [m * 256 * 16 B] (bf16 codebook),
 *                                 hx_out [ntotal] floats, pmax2_out [nlist] floats.
 *   eioku_ivfpq_lists_workspace : bytes of device workspace one search of nq queries needs (-1: bad argument).
 *   eioku_ivfpq_search_lists    : the whole search; probes [nq][nprobe] from the coarse quantiser; cand_cap = per-query
 *                                 candidate capacity (0: 8192); a query whose list overflows is redone by the gated
 *                                 query-major scan (same results, slower).  stats_out (optional, device, 6 ints): overflow
 *                                 bits (1: every query redone, 2: some), work items, largest per-query candidate list,
 *                                 candidates of all queries, the query holding the largest list, its size.
 * Replaces: FAISS IndexIVFPQ::search as planned by /root/reference/.kiro/specs/semantic-video-search/design.md:35-40,1105-1113. */
int eioku_ivfpq_lists_aux(const uint8_t* list_codes_dev, const int* offsets_dev, const int* sizes_dev, int nlist, int d, int m,
                          const float* list_tables_dev, const float* pq_dev, void* pqh_out_dev, float* hx_out_dev,
                          float* pmax2_out_dev, void* stream);
long long eioku_ivfpq_lists_workspace(int nq, int d, int nprobe, int nlist, long long ntotal, int k, int cand_cap);
int eioku_ivfpq_search_lists(const float* q_dev, int nq, int d, int m, const long long* probes_dev, int nprobe, int nlist,
                             long long ntotal, const float* coarse_dev, const float* pq_dev, const int* offsets_dev, const int* sizes_dev,
                             const uint8_t* list_codes_dev, const long long* list_ids_dev, const float* list_tables_dev,
                             const float* query_tables_dev, const void* pqh_dev, const float* hx_dev,
                             const float* pmax2_dev, int k, int cand_cap, void* workspace_dev, long long workspace_bytes,
                             float* D_dev, long long* I_dev, int* stats_out_dev, void* stream);

/* ---- place classification: Places365 ResNet18 (csrc/resnet.hip) --------------------------------------------------
 * Replaces the per-frame arithmetic of ModelManager.classify_places
 * (/root/reference/ml-service/src/services/model_manager.py:560-713): `transforms.Resize((224, 224))` on the PIL image
 * (Pillow's antialiased bilinear resample, bit-exact: model_manager.py:630-640), ToTensor + Normalize,
 * `models.resnet18` with a 365-way fc (:609-624), softmax + descending sort + top_k (:672-687).
 * Weights are set per convolution with BatchNorm folded in (eioku_resnet18_conv_info names them as torchvision's
 * state dict does: "conv1", "layer1.0.conv1", ..., "layer2.0.downsample.0"); fp16 storage, fp32 accumulation.
 * The resize tables are Pillow's precompute_coeffs output for (w -> 224) and (h -> 224), computed by the host
 * (eioku_amd/places.py): bounds [224][2] = {first input index, taps}, k [224][ksize] int32 taps x 2^22. */
typedef struct eioku_resnet eioku_resnet_t;
int eioku_resnet18_create(int num_classes, eioku_resnet_t** out);
void eioku_resnet18_destroy(eioku_resnet_t* r);
int eioku_resnet18_num_convs(const eioku_resnet_t* r);
int eioku_resnet18_conv_info(const eioku_resnet_t* r, int idx, char* name, size_t name_cap, int* cout, int* cin, int* ksize,
                             int* stride);
int eioku_resnet18_set_conv(eioku_resnet_t* r, int idx, const float* weight_oihw, const float* bias);
int eioku_resnet18_set_fc(eioku_resnet_t* r, const float* weight, const float* bias);
/* resize + normalise only: out_f16 [n][224][224][4] fp16 (R, G, B, 0) and / or out_rgb_u8 [n][224][224][3], DEVICE */
int eioku_places_preprocess(eioku_resnet_t* r, const uint8_t* bgr, int n, int h, int w, const int32_t* xbounds,
                            const int32_t* xk, int kx, const int32_t* ybounds, const int32_t* yk, int ky, void* out_f16,
                            uint8_t* out_rgb_u8, int mem, void* stream);
/* the raw network: in [n][224][224][4] fp16 -> logits [n][num_classes] fp32; device pointers, asynchronous */
int eioku_resnet18_forward(eioku_resnet_t* r, const void* in_nhwc4_f16, int n, float* logits_out, void* stream);
/* n BGR frames (host or device) -> per frame the top_k (probability, class) of softmax(logits) in descending order
 * (ties: lower class); outputs on the side `mem` names (host outputs synchronise); logits_out optional */
int eioku_resnet18_classify(eioku_resnet_t* r, const uint8_t* bgr, int n, int h, int w, const int32_t* xbounds,
                            const int32_t* xk, int kx, const int32_t* ybounds, const int32_t* yk, int ky, int top_k,
                            float* prob_out, int32_t* class_out, float* logits_out, int mem, void* stream);
int eioku_resnet18_last_flops(const eioku_resnet_t* r, double* flops);

#ifdef __cplusplus
}
#endif
#endif /* EIOKU_HIP_H */
