// YOLOv8 detection runner: builds the layer plan of the Ultralytics yolov8{n,s,m,l,x}[-face] graph
// over NHWC fp16 buffers (concat / chunk expressed as channel slices, never copied) and runs
// letterbox -> backbone/neck/head convs (K4) -> decode (K6) -> NMS (K7) on one HIP stream.
//
// Graph (ultralytics/cfg/models/v8/yolov8.yaml; ch = backbone widths c1..c5, d = C2f repeats):
//   0 Conv(3,c1,3,2) 1 Conv(c1,c2,3,2) 2 C2f(c2,c2,d0,sc) 3 Conv(c2,c3,3,2) 4 C2f(c3,c3,d1,sc)
//   5 Conv(c3,c4,3,2) 6 C2f(c4,c4,d2,sc) 7 Conv(c4,c5,3,2) 8 C2f(c5,c5,d3,sc) 9 SPPF(c5,c5,5)
//   10 Up 11 Cat(10,6) 12 C2f(c5+c4,c4,d0) 13 Up 14 Cat(13,4) 15 C2f(c4+c3,c3,d0)
//   16 Conv(c3,c3,3,2) 17 Cat(16,12) 18 C2f(c3+c4,c4,d0) 19 Conv(c4,c4,3,2) 20 Cat(19,9)
//   21 C2f(c4+c5,c5,d0) 22 Detect(nc; P3=15,P4=18,P5=21)
#include <algorithm>
#include <array>
#include <memory>
#include <string>
#include <vector>

#include "yolo_ops.h"

#include <cstdlib>

using namespace eioku;

namespace {

constexpr int kRegMax = 16;

struct Buf {
  int level;  // 0 = input resolution, k = H / 2^k
  int ch;
  __half* ptr = nullptr;
  size_t cap = 0;  // bytes
};

enum OpKind { kConv, kPool, kUp };

struct Op {
  OpKind kind;
  int conv = -1;  // weight index
  int in_buf, in_off, in_ch;
  int out_buf, out_off;
  int res_buf = -1, res_off = 0;
  int act = kActSiLU;
  int f32_out = -1;  // index into head outputs (0..5) or -1
  bool chain_next = false;     // this op's output is read by the NEXT op only, before anything overwrites it (C2f's shared tmp)
  bool sole_consumer = false;  // the NEXT op is the only reader of this op's output buffer (set by build_graph)
  int up_consumer = -1;        // kUp: index of the 1x1 conv that can read the low-resolution source in place
  int up_from = -1;            // conv: index of the kUp op whose output slice it reads (then that op is skipped)
};

}  // namespace

struct eioku_yolo {
  int ch[5], depth[4], nc;
  std::vector<std::string> names;
  std::vector<std::array<int, 4>> shapes;  // cout, cin, k, stride
  std::vector<ConvWeights> weights;
  std::vector<bool> set;
  std::vector<Buf> bufs;
  std::vector<Op> ops;
  int in_buf = -1;
  int head_src[3];
  // per-shape workspace
  int cur_n = 0, cur_h = 0, cur_w = 0;
  float* head[6] = {};  // box P3,P4,P5 ; cls P3,P4,P5 (fp32)
  size_t head_cap[6] = {};
  unsigned long long* clsmax[3] = {};  // per anchor (argmax << 32 | max-logit bits): detect()'s class branch output
  size_t clsmax_cap[3] = {};
  int32_t* lvl = nullptr;  // lazy box branch: [N][A] per-level lists of passing anchors, then [N][3] counts
  size_t lvl_cap = 0;
  // deep lazy box branch: flat pixel lists [N*A] (anchors) + [9*N*A] (their neighbourhoods) + 6 counters, and the
  // share of anchors that passed in the last finished call (pinned host copy, written by the GPU, read without a
  // sync: it only steers lazy vs dense evaluation of the branch's 3x3 layers -- both give the same bytes)
  int32_t* flat = nullptr;
  size_t flat_cap = 0;
  int32_t* pass_host = nullptr;  // [4]: cnt1[3], N*A of that call
  hipEvent_t pass_event = nullptr;  // recorded behind the copy into pass_host; the word is read only once it has fired
  bool pass_pending = false;
  long long pass_seen[2] = {0, 0};  // last completed {passed, total}: what the decision uses until newer history lands
  int deep_idx[3][2] = {{-1, -1}, {-1, -1}, {-1, -1}};  // op indices of cv2.l.0 / cv2.l.1
  Cand* cands = nullptr;  // dense [N][A] followed by keys [N][A]
  size_t cands_cap = 0;
  int32_t* counts = nullptr;  // [N] cand counts + [N] det counts
  size_t counts_cap = 0;
  Det* dets = nullptr;
  size_t dets_cap = 0;
  // letterbox tables
  void* lb_tables = nullptr;
  size_t lb_cap = 0;
  unsigned long long lb_key = 0;   // FNV-1a of the bilinear tables currently resident in lb_tables
  void* lb_key_ptr = nullptr;
  __half* lb_out = nullptr;  // == bufs[in_buf].ptr
  double conv_flops_last = 0;
};

namespace {

int add_buf(eioku_yolo* y, int level, int ch) {
  y->bufs.push_back(Buf{level, ch});
  return (int)y->bufs.size() - 1;
}

int add_conv_w(eioku_yolo* y, const std::string& name, int cout, int cin, int k, int s) {
  y->names.push_back(name);
  y->shapes.push_back({cout, cin, k, s});
  return (int)y->names.size() - 1;
}

void add_conv(eioku_yolo* y, const std::string& name, int cin, int cout, int k, int s, int in_buf, int in_off,
              int out_buf, int out_off, int res_buf = -1, int res_off = 0, int act = kActSiLU, int f32_out = -1) {
  Op op;
  op.kind = kConv;
  op.conv = add_conv_w(y, name, cout, cin, k, s);
  op.in_buf = in_buf;
  op.in_off = in_off;
  op.in_ch = cin;
  op.out_buf = out_buf;
  op.out_off = out_off;
  op.res_buf = res_buf;
  op.res_off = res_off;
  op.act = act;
  op.f32_out = f32_out;
  y->ops.push_back(op);
}

// C2f(c1 -> c2, n bottlenecks, shortcut): input slice (in_buf,in_off,c1) -> output slice (out_buf,out_off)
void add_c2f(eioku_yolo* y, const std::string& p, int level, int c1, int c2, int n, bool shortcut, int in_buf,
             int in_off, int out_buf, int out_off) {
  const int c = c2 / 2;
  const int cat = add_buf(y, level, (2 + n) * c);
  const int tmp = add_buf(y, level, c);
  add_conv(y, p + ".cv1.conv", c1, 2 * c, 1, 1, in_buf, in_off, cat, 0);
  for (int i = 0; i < n; ++i) {
    const std::string m = p + ".m." + std::to_string(i);
    add_conv(y, m + ".cv1.conv", c, c, 3, 1, cat, (1 + i) * c, tmp, 0);
    add_conv(y, m + ".cv2.conv", c, c, 3, 1, tmp, 0, cat, (2 + i) * c, shortcut ? cat : -1, (1 + i) * c);
  }
  add_conv(y, p + ".cv2.conv", (2 + n) * c, c2, 1, 1, cat, 0, out_buf, out_off);
}

void build_graph(eioku_yolo* y) {
  const int c1 = y->ch[0], c2 = y->ch[1], c3 = y->ch[2], c4 = y->ch[3], c5 = y->ch[4];
  const int d0 = y->depth[0], d1 = y->depth[1], d2 = y->depth[2], d3 = y->depth[3];
  y->in_buf = add_buf(y, 0, 8);  // RGB + 5 zero channels (16-byte pixel)
  const int t0 = add_buf(y, 1, c1), t1 = add_buf(y, 2, c2), t2 = add_buf(y, 2, c2);
  const int t3 = add_buf(y, 3, c3), t5 = add_buf(y, 4, c4), t7 = add_buf(y, 5, c5), t8 = add_buf(y, 5, c5);
  const int cat14 = add_buf(y, 3, c4 + c3);  // [up(12) | out4]
  const int cat11 = add_buf(y, 4, c5 + c4);  // [up(9)  | out6]
  const int cat20 = add_buf(y, 5, c4 + c5);  // [conv19 | out9]
  const int cat17 = add_buf(y, 4, c3 + c4);  // [conv16 | out12]
  const int sppf = add_buf(y, 5, 2 * c5);    // [cv1 | m1 | m2 | m3], c5/2 each
  const int t15 = add_buf(y, 3, c3), t18 = add_buf(y, 4, c4), t21 = add_buf(y, 5, c5);

  add_conv(y, "model.0.conv", 8, c1, 3, 2, y->in_buf, 0, t0, 0);
  add_conv(y, "model.1.conv", c1, c2, 3, 2, t0, 0, t1, 0);
  add_c2f(y, "model.2", 2, c2, c2, d0, true, t1, 0, t2, 0);
  add_conv(y, "model.3.conv", c2, c3, 3, 2, t2, 0, t3, 0);
  add_c2f(y, "model.4", 3, c3, c3, d1, true, t3, 0, cat14, c4);
  add_conv(y, "model.5.conv", c3, c4, 3, 2, cat14, c4, t5, 0);
  add_c2f(y, "model.6", 4, c4, c4, d2, true, t5, 0, cat11, c5);
  add_conv(y, "model.7.conv", c4, c5, 3, 2, cat11, c5, t7, 0);
  add_c2f(y, "model.8", 5, c5, c5, d3, true, t7, 0, t8, 0);
  // SPPF
  const int ch = c5 / 2;
  add_conv(y, "model.9.cv1.conv", c5, ch, 1, 1, t8, 0, sppf, 0);
  for (int i = 0; i < 3; ++i) {
    Op op;
    op.kind = kPool;
    op.in_buf = sppf;
    op.in_off = i * ch;
    op.in_ch = ch;
    op.out_buf = sppf;
    op.out_off = (i + 1) * ch;
    y->ops.push_back(op);
  }
  add_conv(y, "model.9.cv2.conv", 4 * ch, c5, 1, 1, sppf, 0, cat20, c4);
  // neck
  auto add_up = [&](int in_buf, int in_off, int chn, int out_buf, int out_off) {
    Op op;
    op.kind = kUp;
    op.in_buf = in_buf;
    op.in_off = in_off;
    op.in_ch = chn;
    op.out_buf = out_buf;
    op.out_off = out_off;
    y->ops.push_back(op);
  };
  add_up(cat20, c4, c5, cat11, 0);
  add_c2f(y, "model.12", 4, c5 + c4, c4, d0, false, cat11, 0, cat17, c3);
  add_up(cat17, c3, c4, cat14, 0);
  add_c2f(y, "model.15", 3, c4 + c3, c3, d0, false, cat14, 0, t15, 0);
  add_conv(y, "model.16.conv", c3, c3, 3, 2, t15, 0, cat17, 0);
  add_c2f(y, "model.18", 4, c3 + c4, c4, d0, false, cat17, 0, t18, 0);
  add_conv(y, "model.19.conv", c4, c4, 3, 2, t18, 0, cat20, 0);
  add_c2f(y, "model.21", 5, c4 + c5, c5, d0, false, cat20, 0, t21, 0);
  // Detect
  const int cb = std::max(std::max(16, c3 / 4), kRegMax * 4);
  const int cc = std::max(c3, std::min(y->nc, 100));
  const int src[3] = {t15, t18, t21};
  const int srcc[3] = {c3, c4, c5};
  for (int l = 0; l < 3; ++l) {
    y->head_src[l] = src[l];
    const int lvl = 3 + l;
    const std::string i = std::to_string(l);
    const int b1 = add_buf(y, lvl, cb), b2 = add_buf(y, lvl, cb), k1 = add_buf(y, lvl, cc), k2 = add_buf(y, lvl, cc);
    add_conv(y, "model.22.cv2." + i + ".0.conv", srcc[l], cb, 3, 1, src[l], 0, b1, 0);
    add_conv(y, "model.22.cv2." + i + ".1.conv", cb, cb, 3, 1, b1, 0, b2, 0);
    add_conv(y, "model.22.cv2." + i + ".2", cb, 4 * kRegMax, 1, 1, b2, 0, -1, 0, -1, 0, kActNone, l);
    add_conv(y, "model.22.cv3." + i + ".0.conv", srcc[l], cc, 3, 1, src[l], 0, k1, 0);
    add_conv(y, "model.22.cv3." + i + ".1.conv", cc, cc, 3, 1, k1, 0, k2, 0);
    add_conv(y, "model.22.cv3." + i + ".2", cc, y->nc, 1, 1, k2, 0, -1, 0, -1, 0, kActNone, 3 + l);
  }
  y->weights.resize(y->names.size());
  y->set.assign(y->names.size(), false);
  // box branch per level: cv2.l.0 -> cv2.l.1 -> cv2.l.2 (f32_out = l); remember the two 3x3 ops
  for (size_t i = 0; i < y->ops.size(); ++i) {
    const Op& o2 = y->ops[i];
    if (o2.kind != kConv || o2.f32_out < 0 || o2.f32_out >= 3) continue;
    int i1 = -1, i0 = -1;
    for (size_t j = 0; j < y->ops.size(); ++j)
      if (y->ops[j].kind == kConv && y->ops[j].f32_out < 0 && y->ops[j].out_buf == o2.in_buf) i1 = (int)j;
    if (i1 >= 0)
      for (size_t j = 0; j < y->ops.size(); ++j)
        if (y->ops[j].kind == kConv && y->ops[j].f32_out < 0 && y->ops[j].out_buf == y->ops[i1].in_buf) i0 = (int)j;
    y->deep_idx[o2.f32_out][0] = i0;
    y->deep_idx[o2.f32_out][1] = i1;
  }
  // upsample -> concat -> 1x1: the conv reads the half-resolution source itself when it is the only reader of
  // the upsampled slice (which sits at channel 0 of the concat buffer)
  for (size_t i = 0; i < y->ops.size(); ++i) {
    Op& up = y->ops[i];
    if (up.kind != kUp || up.out_off != 0) continue;
    int readers = 0, consumer = -1;
    for (size_t j = 0; j < y->ops.size(); ++j) {
      const Op& o = y->ops[j];
      if (o.in_buf == up.out_buf && o.in_off < up.in_ch) {
        ++readers;
        consumer = (int)j;
      }
      if (o.res_buf == up.out_buf) readers += 2;
    }
    if (readers != 1 || consumer < (int)i) continue;
    Op& c = y->ops[consumer];
    if (c.kind == kConv && y->shapes[c.conv][2] == 1 && c.in_off == 0 && c.in_ch >= up.in_ch && up.in_ch % 32 == 0) {
      up.up_consumer = consumer;
      c.up_from = (int)i;
    }
  }
  // an op whose whole output buffer is read by the next op and by nothing else may hand its tile over on chip
  for (size_t i = 0; i + 1 < y->ops.size(); ++i) {
    Op& op = y->ops[i];
    if (op.kind != kConv || op.f32_out >= 0 || op.out_buf == y->in_buf) continue;
    int readers = 0;
    for (size_t j = 0; j < y->ops.size(); ++j)
      readers += (y->ops[j].in_buf == op.out_buf) + (y->ops[j].res_buf == op.out_buf);
    const Op& nx = y->ops[i + 1];
    int writers = 0;
    for (const Op& o : y->ops) writers += (o.f32_out < 0 && o.out_buf == op.out_buf);
    op.sole_consumer = readers == 1 && writers == 1 && nx.in_buf == op.out_buf && nx.res_buf != op.out_buf &&
                       y->bufs[op.out_buf].ch == y->shapes[op.conv][0];
    // weaker than sole_consumer: the buffer may be reused (C2f's tmp is shared by its bottlenecks) as long as every
    // read of it is the op right after the write it consumes
    bool priv = nx.in_buf == op.out_buf && nx.in_off == op.out_off && y->bufs[op.out_buf].ch == y->shapes[op.conv][0];
    for (size_t j = 0; j < y->ops.size() && priv; ++j) {
      const Op& o = y->ops[j];
      if (o.res_buf == op.out_buf) priv = false;
      if (o.in_buf == op.out_buf) {
        const Op* pv = j > 0 ? &y->ops[j - 1] : nullptr;
        priv = priv && pv && pv->kind == kConv && pv->f32_out < 0 && pv->out_buf == op.out_buf && pv->out_off == o.in_off;
      }
    }
    op.chain_next = priv;
  }
}

int level_dim(int x, int level) {
  for (int i = 0; i < level; ++i) x = conv_out_dim(x, 3, 2);
  return x;
}

template <typename T>
int ensure(T** p, size_t* cap, size_t bytes) {
  if (*cap >= bytes) return EIOKU_OK;
  if (*p) {
    (void)hipDeviceSynchronize();
    (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
  }
  EIOKU_HIP_CHECK(hipMalloc((void**)p, bytes));
  *cap = bytes;
  return EIOKU_OK;
}

int prepare(eioku_yolo* y, int n, int h, int w) {
  EIOKU_REQUIRE(h % 32 == 0 && w % 32 == 0, "network input %dx%d must be a multiple of 32", h, w);
  // the conv kernels address activations with 32-bit element offsets and decode tile / pixel indices below 2^24
  // (fast_div): a batch that breaks either must be split by the caller -- refuse it loudly
  for (const auto& b : y->bufs) {
    const long long px = (long long)n * level_dim(h, b.level) * level_dim(w, b.level);
    EIOKU_REQUIRE(px * b.ch < (1ll << 31) && px < (1ll << 24) * 16,
                  "batch of %d frames at %dx%d: an activation tensor of %lld x %d channels exceeds one launch's 32-bit "
                  "offsets -- split the batch", n, h, w, px, b.ch);
  }
  for (auto& b : y->bufs) {
    const size_t bytes = (size_t)n * level_dim(h, b.level) * level_dim(w, b.level) * b.ch * sizeof(__half);
    int rc = ensure(&b.ptr, &b.cap, bytes);
    if (rc) return rc;
  }
  int A = 0;
  for (int l = 0; l < 3; ++l) {
    const size_t px = (size_t)n * level_dim(h, 3 + l) * level_dim(w, 3 + l);
    A += level_dim(h, 3 + l) * level_dim(w, 3 + l);
    int rc = ensure(&y->head[l], &y->head_cap[l], px * 4 * kRegMax * sizeof(float));
    if (rc) return rc;
    rc = ensure(&y->head[3 + l], &y->head_cap[3 + l], px * y->nc * sizeof(float));
    if (rc) return rc;
    rc = ensure(&y->clsmax[l], &y->clsmax_cap[l], px * sizeof(unsigned long long));
    if (rc) return rc;
  }
  int rc = ensure(&y->cands, &y->cands_cap, (size_t)n * A * (sizeof(Cand) + sizeof(unsigned long long)));
  if (rc) return rc;
  rc = ensure(&y->counts, &y->counts_cap, (size_t)n * 2 * sizeof(int32_t));
  if (rc) return rc;
  rc = ensure(&y->lvl, &y->lvl_cap, ((size_t)n * A + (size_t)n * 3) * sizeof(int32_t));
  if (rc) return rc;
  rc = ensure(&y->flat, &y->flat_cap, ((size_t)10 * n * A + 16) * sizeof(int32_t));
  if (rc) return rc;
  y->cur_n = n;
  y->cur_h = h;
  y->cur_w = w;
  return EIOKU_OK;
}

// total of the per-image candidate counts and the anchor total of this call -> 4 ints, copied to a pinned host
// word without a sync; the NEXT call reads whatever has arrived (see the deep lazy box branch)
__global__ void k_pass_record(const int32_t* counts, int n, int total, int32_t* out) {
  __shared__ int s;
  if (threadIdx.x == 0) s = 0;
  __syncthreads();
  int v = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) v += counts[i];
  atomicAdd(&s, v);
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = s;
    out[1] = 0;
    out[2] = 0;
    out[3] = total;
  }
}

int record_pass_rate(eioku_yolo* y, int n, int A, hipStream_t stream) {
  int32_t* dev = y->flat + (size_t)10 * n * A + 8;
  hipLaunchKernelGGL(k_pass_record, dim3(1), dim3(256), 0, stream, y->counts, n, n * A, dev);
  EIOKU_HIP_CHECK(hipMemcpyAsync(y->pass_host, dev, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  EIOKU_HIP_CHECK(hipEventRecord(y->pass_event, stream));
  y->pass_pending = true;
  return EIOKU_OK;
}

int run_ops(eioku_yolo* y, int n, int h, int w, hipStream_t stream, const FusedInput* fused, double* flops_out,
            bool clsmax = false, bool lazybox = false, bool lazydeep = false) {
  double flops = 0;
  bool skip_next = false;
  int pool_skip = 0, front_skip = 0;
  for (const Op& op : y->ops) {
    if (front_skip > 0) {  // ran inside the fused front-end launch
      --front_skip;
      continue;
    }
    if (skip_next) {  // this 1x1 ran inside the previous launch
      skip_next = false;
      continue;
    }
    const Buf& ib = y->bufs[op.in_buf];
    const int H = level_dim(h, ib.level), W = level_dim(w, ib.level);
    Slice in{ib.ptr, ib.ch, op.in_off};
    int rc = EIOKU_OK;
    if (op.kind == kConv) {
      const ConvWeights& cw = y->weights[op.conv];
      Slice out{}, res{};
      float* f32 = nullptr;
      if (op.f32_out >= 0) {
        f32 = y->head[op.f32_out];
      } else {
        const Buf& ob = y->bufs[op.out_buf];
        out = Slice{ob.ptr, ob.ch, op.out_off};
      }
      if (op.res_buf >= 0) res = Slice{y->bufs[op.res_buf].ptr, y->bufs[op.res_buf].ch, op.res_off};
      const bool first = &op == &y->ops.front();
      // 3x3 whose output feeds only the next op, a 1x1 (stride-2 conv -> C2f.cv1): one launch, no intermediate
      const Op* nx = (&op + 1 <= &y->ops.back()) ? &op + 1 : nullptr;
      const bool pair = !first && nx && nx->kind == kConv && op.f32_out < 0 && op.res_buf < 0 && nx->res_buf < 0 &&
                        nx->f32_out < 0 && nx->in_buf == op.out_buf && nx->in_off == op.out_off && op.sole_consumer &&
                        conv_post_ok(cw, y->weights[nx->conv]);
      // fused-letterbox stem -> model.1 -> model.2.cv1 (YOLOv8n, copy-mode frames): one launch
      if (first && fused && y->ops.size() > 2) {
        const Op& o1 = y->ops[1];
        const Op& o2 = y->ops[2];
        const bool front = o1.kind == kConv && o2.kind == kConv && op.f32_out < 0 && op.res_buf < 0 && o1.f32_out < 0 &&
                           o1.res_buf < 0 && o2.f32_out < 0 && o2.res_buf < 0 && o1.in_buf == op.out_buf &&
                           o1.in_off == op.out_off && op.sole_consumer && o2.in_buf == o1.out_buf && o2.in_off == o1.out_off &&
                           o1.sole_consumer &&
                           conv_stem_chain_ok(cw, y->weights[o1.conv], y->weights[o2.conv], *fused, W);
        if (front) {
          const ConvWeights& w1 = y->weights[o1.conv];
          const ConvWeights& w2 = y->weights[o2.conv];
          const Buf& ob2 = y->bufs[o2.out_buf];
          rc = conv_stem_chain_forward(cw, w1, w2, *fused, n, H, W, Slice{ob2.ptr, ob2.ch, o2.out_off}, op.act, o1.act, o2.act,
                                       stream);
          const int H1 = conv_out_dim(H, 3, 2), W1 = conv_out_dim(W, 3, 2);
          const double px2 = (double)n * conv_out_dim(H1, 3, 2) * conv_out_dim(W1, 3, 2);
          flops += cw.flops_per_pixel() * n * H1 * W1 + (w1.flops_per_pixel() + w2.flops_per_pixel()) * px2;
          front_skip = 2;
          if (rc) return rc;
          continue;
        }
      }
      // C2f Bottleneck (3x3 -> 3x3 [+ x]) on the shallow levels: one launch, the intermediate stays in LDS
      const bool chain = !first && nx && nx->kind == kConv && op.f32_out < 0 && op.res_buf < 0 && nx->f32_out < 0 &&
                         op.chain_next && conv_chain_ok(cw, y->weights[nx->conv]) &&
                         (nx->res_buf < 0 || (nx->res_buf == op.in_buf && nx->res_off == op.in_off));
      if (chain) {
        const ConvWeights& bw = y->weights[nx->conv];
        const Buf& ob2 = y->bufs[nx->out_buf];
        // ... and the C2f's closing 1x1 when this is its LAST Bottleneck: concat = [y0 | y1 | .. | y_out], y_out = this
        // pair's output (the concat's last slice, read by that 1x1 alone), the pair's input = the slice before it
        const Op* n2 = (&op + 2 <= &y->ops.back()) ? &op + 2 : nullptr;
        bool cat = n2 && n2->kind == kConv && n2->f32_out < 0 && n2->res_buf < 0 && n2->in_buf == nx->out_buf &&
                   n2->in_off == 0 && op.in_buf == nx->out_buf && nx->out_off == op.in_off + cw.cin &&
                   n2->in_ch == nx->out_off + cw.cin && ob2.ch == n2->in_ch && conv_chain_cat_ok(cw, bw, y->weights[n2->conv]);
        for (size_t j = 0; j < y->ops.size() && cat; ++j) {  // nobody else reads y2
          const Op& o = y->ops[j];
          if (&o == n2) continue;
          if (o.in_buf == nx->out_buf && o.in_off + o.in_ch > nx->out_off) cat = false;
          if (o.res_buf == nx->out_buf && o.res_off >= nx->out_off) cat = false;
        }
        if (cat) {
          const ConvWeights& w2 = y->weights[n2->conv];
          const Buf& ob3 = y->bufs[n2->out_buf];
          rc = conv_chain_forward(cw, bw, in, n, H, W, Slice{}, nx->res_buf >= 0, op.act, nx->act, stream, &w2,
                                  Slice{ob2.ptr, ob2.ch, 0}, Slice{ob3.ptr, ob3.ch, n2->out_off}, n2->act);
          flops += (cw.flops_per_pixel() + bw.flops_per_pixel() + w2.flops_per_pixel()) * (double)n * H * W;
          front_skip = 2;
          if (rc) return rc;
          continue;
        }
        rc = conv_chain_forward(cw, bw, in, n, H, W, Slice{ob2.ptr, ob2.ch, nx->out_off}, nx->res_buf >= 0, op.act, nx->act,
                                stream);
        flops += (cw.flops_per_pixel() + bw.flops_per_pixel()) * (double)n * H * W;
        skip_next = true;
        if (rc) return rc;
        continue;
      }
      if (pair) {
        const ConvWeights& pw = y->weights[nx->conv];
        const Buf& ob2 = y->bufs[nx->out_buf];
        rc = conv_forward(cw, in, n, H, W, Slice{ob2.ptr, ob2.ch, nx->out_off}, nullptr, Slice{}, op.act, stream, nullptr,
                          &pw, nx->act);
        const double px = (double)n * conv_out_dim(H, cw.ks, cw.stride) * conv_out_dim(W, cw.ks, cw.stride);
        flops += (cw.flops_per_pixel() + pw.flops_per_pixel()) * px;
        skip_next = true;
        if (rc) return rc;
        continue;
      }
      if (lazybox && op.f32_out >= 0 && op.f32_out < 3) continue;  // box branch's last conv: evaluated by decode, per anchor
      if (lazydeep) {  // ... and so are its two 3x3 layers
        const int oi = (int)(&op - &y->ops.front());
        bool skip = false;
        for (int l = 0; l < 3; ++l) skip = skip || oi == y->deep_idx[l][0] || oi == y->deep_idx[l][1];
        if (skip) continue;
      }
      static const bool up_off = getenv("EIOKU_UP_FUSE") && atoi(getenv("EIOKU_UP_FUSE")) == 0;
      if (op.up_from >= 0 && !up_off && (long long)n * H * W < (1ll << 24)) {
        const Op& uo = y->ops[op.up_from];
        UpSource us{Slice{y->bufs[uo.in_buf].ptr, y->bufs[uo.in_buf].ch, uo.in_off}, uo.in_ch};
        rc = conv_forward(cw, in, n, H, W, out, f32, res, op.act, stream, nullptr, nullptr, kActSiLU, nullptr, &us);
        flops += cw.flops_per_pixel() * n * conv_out_dim(H, cw.ks, cw.stride) * conv_out_dim(W, cw.ks, cw.stride);
        if (rc) return rc;
        continue;
      }
      if (clsmax && op.f32_out >= 3 && conv_clsmax_ok(cw, op.act))  // class branch: max / argmax words, no logit map
        rc = conv_forward(cw, in, n, H, W, Slice{}, nullptr, Slice{}, op.act, stream, nullptr, nullptr, kActNone,
                          y->clsmax[op.f32_out - 3]);
      else
        rc = conv_forward(cw, in, n, H, W, out, f32, res, op.act, stream, first ? fused : nullptr);
      flops += cw.flops_per_pixel() * n * conv_out_dim(H, cw.ks, cw.stride) * conv_out_dim(W, cw.ks, cw.stride);
    } else if (op.kind == kPool) {
      const Buf& ob = y->bufs[op.out_buf];
      // SPPF: pool -> pool -> pool, each reading the previous one's slice: one launch when the plane fits in LDS
      const Op* p1 = &op + 1 <= &y->ops.back() ? &op + 1 : nullptr;
      const Op* p2 = &op + 2 <= &y->ops.back() ? &op + 2 : nullptr;
      const int step = op.out_off - op.in_off;
      const bool chain = pool_skip == 0 && p1 && p2 && p1->kind == kPool && p2->kind == kPool && p1->in_buf == op.out_buf &&
                         p2->in_buf == op.out_buf && p1->out_buf == op.out_buf && p2->out_buf == op.out_buf &&
                         p1->in_off == op.out_off && p2->in_off == p1->out_off && p1->out_off - p1->in_off == step &&
                         p2->out_off - p2->in_off == step && p1->in_ch == op.in_ch && p2->in_ch == op.in_ch &&
                         op.in_buf == op.out_buf && (size_t)H * W * 48 <= 144 * 1024;
      if (pool_skip > 0) {
        --pool_skip;
        continue;
      }
      if (chain) {
        rc = sppf_pools_forward(in, Slice{ob.ptr, ob.ch, op.out_off}, step, n, H, W, op.in_ch, stream);
        pool_skip = 2;
      } else {
        rc = maxpool5_forward(in, Slice{ob.ptr, ob.ch, op.out_off}, n, H, W, op.in_ch, stream);
      }
    } else {
      const Buf& ob = y->bufs[op.out_buf];
      static const bool up_off2 = getenv("EIOKU_UP_FUSE") && atoi(getenv("EIOKU_UP_FUSE")) == 0;
      // the consumer's pixel count (4x this op's) decides, exactly as in the conv branch above
      if (op.up_consumer >= 0 && !up_off2 && (long long)n * H * W * 4 < (1ll << 24)) continue;
      rc = upsample2x_forward(in, Slice{ob.ptr, ob.ch, op.out_off}, n, H, W, op.in_ch, stream);
    }
    if (rc) return rc;
  }
  *flops_out = flops;
  return EIOKU_OK;
}

int run_network(eioku_yolo* y, int n, int h, int w, hipStream_t stream, const FusedInput* fused = nullptr,
                bool clsmax = false, bool lazybox = false, bool lazydeep = false) {
  for (size_t i = 0; i < y->set.size(); ++i)
    EIOKU_REQUIRE(y->set[i], "conv %zu (%s) has no weights", i, y->names[i].c_str());
  // (replaying the forward as a hipGraph was built and measured on ROCm 7.2 / MI355X: 3.07 ms per bench step against
  // 3.03 ms for plain launches -- the gaps between dependent kernels are the GPU's, not the host's -- and removed)
  return run_ops(y, n, h, w, stream, fused, &y->conv_flops_last, clsmax, lazybox, lazydeep);
}

}  // namespace

extern "C" {

int eioku_yolo_create(const int* ch5, const int* depth4, int nc, eioku_yolo** out) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(ch5 && depth4 && out, "NULL argument");
  EIOKU_REQUIRE(nc >= 1 && nc <= 1024, "nc %d out of range", nc);
  for (int i = 0; i < 5; ++i) EIOKU_REQUIRE(ch5[i] >= 16 && ch5[i] % 16 == 0, "width ch[%d]=%d must be a multiple of 16", i, ch5[i]);
  for (int i = 0; i < 4; ++i) EIOKU_REQUIRE(depth4[i] >= 1 && depth4[i] <= 12, "depth[%d]=%d out of range", i, depth4[i]);
  auto* y = new eioku_yolo();
  for (int i = 0; i < 5; ++i) y->ch[i] = ch5[i];
  for (int i = 0; i < 4; ++i) y->depth[i] = depth4[i];
  y->nc = nc;
  build_graph(y);
  *out = y;
  return EIOKU_OK;
}

void eioku_yolo_destroy(eioku_yolo* y) {
  if (!y) return;
  (void)hipDeviceSynchronize();
  for (auto& w : y->weights) conv_weights_destroy(&w);
  for (auto& b : y->bufs)
    if (b.ptr) (void)hipFree(b.ptr);
  for (int i = 0; i < 6; ++i)
    if (y->head[i]) (void)hipFree(y->head[i]);
  for (int i = 0; i < 3; ++i)
    if (y->clsmax[i]) (void)hipFree(y->clsmax[i]);
  if (y->lvl) (void)hipFree(y->lvl);
  if (y->flat) (void)hipFree(y->flat);
  if (y->pass_host) (void)hipHostFree(y->pass_host);
  if (y->pass_event) (void)hipEventDestroy(y->pass_event);
  if (y->cands) (void)hipFree(y->cands);
  if (y->counts) (void)hipFree(y->counts);
  if (y->dets) (void)hipFree(y->dets);
  if (y->lb_tables) (void)hipFree(y->lb_tables);
  delete y;
}

int eioku_yolo_num_convs(const eioku_yolo* y) { return y ? (int)y->names.size() : 0; }

int eioku_yolo_conv_info(const eioku_yolo* y, int idx, char* name, size_t cap, int* cout, int* cin, int* k,
                         int* stride) {
  EIOKU_REQUIRE(y && idx >= 0 && idx < (int)y->names.size(), "bad conv index %d", idx);
  if (name && cap) snprintf(name, cap, "%s", y->names[idx].c_str());
  if (cout) *cout = y->shapes[idx][0];
  if (cin) *cin = y->shapes[idx][1];
  if (k) *k = y->shapes[idx][2];
  if (stride) *stride = y->shapes[idx][3];
  return EIOKU_OK;
}

int eioku_yolo_set_conv(eioku_yolo* y, int idx, const float* w_oihw, const float* bias) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(y && idx >= 0 && idx < (int)y->names.size(), "bad conv index %d", idx);
  EIOKU_REQUIRE(w_oihw, "NULL weights");
  const auto& s = y->shapes[idx];
  if (y->set[idx]) conv_weights_destroy(&y->weights[idx]);
  int rc = conv_weights_create(&y->weights[idx], s[0], s[1], s[2], s[3], w_oihw, bias);
  y->set[idx] = rc == EIOKU_OK;
  return rc;
}

int eioku_yolo_forward(eioku_yolo* y, const void* in_nhwc8, int n, int h, int w, float* const* box_out,
                       float* const* cls_out, void* stream_) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(y && in_nhwc8, "NULL argument");
  hipStream_t stream = (hipStream_t)stream_;
  int rc = prepare(y, n, h, w);
  if (rc) return rc;
  EIOKU_HIP_CHECK(hipMemcpyAsync(y->bufs[y->in_buf].ptr, in_nhwc8, (size_t)n * h * w * 8 * sizeof(__half),
                                 hipMemcpyDeviceToDevice, stream));
  rc = run_network(y, n, h, w, stream);
  if (rc) return rc;
  for (int l = 0; l < 3; ++l) {
    const size_t px = (size_t)n * level_dim(h, 3 + l) * level_dim(w, 3 + l);
    if (box_out && box_out[l])
      EIOKU_HIP_CHECK(hipMemcpyAsync(box_out[l], y->head[l], px * 64 * sizeof(float), hipMemcpyDeviceToDevice, stream));
    if (cls_out && cls_out[l])
      EIOKU_HIP_CHECK(hipMemcpyAsync(cls_out[l], y->head[3 + l], px * y->nc * sizeof(float), hipMemcpyDeviceToDevice, stream));
  }
  return EIOKU_OK;
}

int eioku_yolo_last_conv_flops(const eioku_yolo* y, double* flops) {
  EIOKU_REQUIRE(y && flops, "NULL argument");
  *flops = y->conv_flops_last;
  return EIOKU_OK;
}

// Detection from raw frames.  The letterbox tables (xofs/yofs/xalpha/ybeta and the geometry) are
// computed by the host in the reference's own Python float semantics and passed in `lb`:
//   lb_geom[0..8] = {new_h, new_w, top, left, out_h, out_w, mode, pad_x, pad_y}
// (pad_x/pad_y are scale_boxes' own rounding of the padding, which need not equal left/top).
int eioku_yolo_detect(eioku_yolo* y, const uint8_t* bgr, int n, int h, int w, const int32_t* lb_geom,
                      const int32_t* xofs, const int32_t* yofs, const int16_t* xalpha, const int16_t* ybeta,
                      float gain, float conf, float iou, int max_det, void* dets_out, int32_t* counts_out,
                      int mem, void* stream_) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(y && lb_geom, "NULL argument");
  EIOKU_REQUIRE(mem == EIOKU_MEM_HOST || mem == EIOKU_MEM_DEVICE, "bad mem flag %d", mem);
  EIOKU_REQUIRE(n >= 0 && h > 0 && w > 0, "bad frame shape");
  if (n == 0) return EIOKU_OK;
  EIOKU_REQUIRE(bgr && dets_out && counts_out, "NULL buffer");
  hipStream_t stream = (hipStream_t)stream_;
  LetterboxPlan p;
  p.src_h = h;
  p.src_w = w;
  p.new_h = lb_geom[0];
  p.new_w = lb_geom[1];
  p.top = lb_geom[2];
  p.left = lb_geom[3];
  p.out_h = lb_geom[4];
  p.out_w = lb_geom[5];
  p.mode = lb_geom[6];
  EIOKU_REQUIRE(p.mode >= 0 && p.mode <= 2, "bad letterbox mode %d", p.mode);
  EIOKU_REQUIRE(p.new_h > 0 && p.new_w > 0 && p.top >= 0 && p.left >= 0 && p.top + p.new_h <= p.out_h &&
                    p.left + p.new_w <= p.out_w, "inconsistent letterbox geometry");
  EIOKU_REQUIRE(p.mode != 1 || (xofs && yofs && xalpha && ybeta), "bilinear letterbox needs its tables");
  EIOKU_REQUIRE(p.mode != 0 || (p.new_h == h && p.new_w == w), "copy mode needs new size == source size");
  EIOKU_REQUIRE(p.mode != 2 || (p.new_h * 2 == h && p.new_w * 2 == w), "area mode needs an exact 1/2 scale");
  int rc = prepare(y, n, p.out_h, p.out_w);
  if (rc) return rc;
  // tables -> device
  const size_t tb = (size_t)p.new_w * 4 + (size_t)p.new_h * 4 + (size_t)p.new_w * 4 + (size_t)p.new_h * 4;
  rc = ensure(&y->lb_tables, &y->lb_cap, tb + 64);
  if (rc) return rc;
  char* t = (char*)y->lb_tables;
  p.xofs = (const int32_t*)t;
  p.yofs = (const int32_t*)(t + (size_t)p.new_w * 4);
  p.xalpha = (const int16_t*)(t + (size_t)p.new_w * 4 + (size_t)p.new_h * 4);
  p.ybeta = (const int16_t*)(t + (size_t)p.new_w * 8 + (size_t)p.new_h * 4);
  if (p.mode == 1) {
    for (int i = 0; i < p.new_w; ++i) EIOKU_REQUIRE(xofs[i] >= 0 && xofs[i] < w, "xofs[%d]=%d outside the frame", i, xofs[i]);
    // The tables depend on the frame size only: upload them once per (geometry, content) -- four pageable H2D
    // copies per call stall the stream on the host and with it the launch-ahead of the whole step.
    unsigned long long key = 1469598103934665603ull;
    auto mix = [&](const void* ptr, size_t bytes) {
      const unsigned char* b = (const unsigned char*)ptr;
      for (size_t i = 0; i < bytes; ++i) key = (key ^ b[i]) * 1099511628211ull;
    };
    const int dims[4] = {h, w, p.new_h, p.new_w};
    mix(dims, sizeof(dims));
    mix(xofs, (size_t)p.new_w * 4);
    mix(yofs, (size_t)p.new_h * 4);
    mix(xalpha, (size_t)p.new_w * 4);
    mix(ybeta, (size_t)p.new_h * 4);
    if (key != y->lb_key || t != y->lb_key_ptr) {
      EIOKU_HIP_CHECK(hipMemcpyAsync((void*)p.xofs, xofs, (size_t)p.new_w * 4, hipMemcpyHostToDevice, stream));
      EIOKU_HIP_CHECK(hipMemcpyAsync((void*)p.yofs, yofs, (size_t)p.new_h * 4, hipMemcpyHostToDevice, stream));
      EIOKU_HIP_CHECK(hipMemcpyAsync((void*)p.xalpha, xalpha, (size_t)p.new_w * 4, hipMemcpyHostToDevice, stream));
      EIOKU_HIP_CHECK(hipMemcpyAsync((void*)p.ybeta, ybeta, (size_t)p.new_h * 4, hipMemcpyHostToDevice, stream));
      EIOKU_HIP_CHECK(hipStreamSynchronize(stream));  // later calls may come on other streams
      y->lb_key = key;
      y->lb_key_ptr = t;
    }
  }
  const uint8_t* d_bgr = bgr;
  if (mem == EIOKU_MEM_HOST) {
    uint8_t* s = (uint8_t*)scratch(kSlotIn, (size_t)n * h * w * 3);
    if (!s) return EIOKU_ENOMEM;
    EIOKU_HIP_CHECK(hipMemcpyAsync(s, bgr, (size_t)n * h * w * 3, hipMemcpyHostToDevice, stream));
    d_bgr = s;
  }
  // copy / exact-half letterbox modes: the stem reads the frames itself (the fp16 network input, 6.5 MB per
  // 640x640 frame, is never written or read); the bilinear mode keeps K3 as its own pass
  // cv2.INTER_LINEAR at an exact odd integer scale k samples source pixel k d + (k - 1) / 2 with tap weights (1, 0) -
  // 1080p -> 360 x 640 is k = 3 - and K3 with such tables returns the source bytes themselves (its fixed-point
  // arithmetic is the identity on weights (2048, 0) x (2048, 0)): the stem can read the frames as a strided copy
  int step = 1, off = 0;
  if (p.mode == 1 && p.new_w > 1 && p.new_h > 1) {
    const int k = xofs[1] - xofs[0];
    bool decim = k >= 2 && yofs[1] - yofs[0] == k && xofs[0] == yofs[0] && xofs[0] >= 0 &&
                 // (the last sampled pixel of a row is followed by another one: the fused front end loads 4 bytes per pixel)
                 (long long)k * (p.new_w - 1) + xofs[0] <= w - 2 && (long long)k * (p.new_h - 1) + yofs[0] < h;
    for (int i = 0; decim && i < p.new_w; ++i) decim = xofs[i] == k * i + xofs[0] && xalpha[2 * i] == 2048 && xalpha[2 * i + 1] == 0;
    for (int j = 0; decim && j < p.new_h; ++j) decim = yofs[j] == k * j + yofs[0] && ybeta[2 * j] == 2048 && ybeta[2 * j + 1] == 0;
    if (decim) {
      step = k;
      off = xofs[0];
    }
  }
  FusedInput fi{d_bgr, p.src_h, p.src_w, p.new_h, p.new_w, p.top, p.left, step > 1 ? 0 : p.mode, step, off};
  const Op& op0 = y->ops.front();
  static const bool no_fuse = getenv("EIOKU_STEM_FUSE") && atoi(getenv("EIOKU_STEM_FUSE")) == 0;
  const bool fuse = !no_fuse && op0.kind == kConv && op0.in_buf == y->in_buf && op0.res_buf < 0 && op0.f32_out < 0 &&
                    fused_input_ok(y->weights[op0.conv], fi, Slice{}, nullptr);
  if (!fuse) {
    rc = letterbox_forward(d_bgr, n, p, y->bufs[y->in_buf].ptr, stream);
    if (rc) return rc;
  }
  // the class branch hands decode {max logit, argmax} per anchor (8 B) instead of the nc-wide fp32 rows, when its
  // last conv can (one cout tile: nc <= 128)
  bool cm = true;
  for (const Op& op : y->ops)
    if (op.kind == kConv && op.f32_out >= 3) cm = cm && conv_clsmax_ok(y->weights[op.conv], op.act);
  // ... and then only the anchors that pass the threshold need their 64 DFL logits: the box branch's last 1x1 conv
  // is evaluated by decode for those (one 64-row weight tile, whole 32-channel chunks)
  static const bool lazy_off = getenv("EIOKU_LAZY_BOX") && atoi(getenv("EIOKU_LAZY_BOX")) == 0;
  bool lazy = cm && !lazy_off;
  LazyBox lbx{};
  for (const Op& op : y->ops)
    if (op.kind == kConv && op.f32_out >= 0 && op.f32_out < 3) {
      const ConvWeights& cw = y->weights[op.conv];
      lazy = lazy && cw.ks == 1 && cw.ntiles == 1 && cw.nf == 4 && cw.cout == 64 && cw.cin % 32 == 0 && op.act == kActNone &&
             op.res_buf < 0 && (lbx.in_cs == 0 || (lbx.in_cs == y->bufs[op.in_buf].ch && lbx.nchunks == cw.nchunks));
      lbx.in[op.f32_out] = y->bufs[op.in_buf].ptr + op.in_off;
      lbx.wgt[op.f32_out] = reinterpret_cast<const uint4*>(cw.d_w);
      lbx.bias[op.f32_out] = cw.d_b;
      lbx.in_cs = y->bufs[op.in_buf].ch;
      lbx.nchunks = cw.nchunks;
    }
  // Deep: with few anchors passing (the usual case at conf >= 0.25) the branch's two 3x3 layers are evaluated only
  // where decode needs them.  Which way to go is decided from the previous call's pass rate (<= 3 %): a wrong guess
  // costs time, never correctness.
  static const float deep_frac = getenv("EIOKU_LAZY_DEEP_FRAC") ? (float)atof(getenv("EIOKU_LAZY_DEEP_FRAC")) : 0.03f;
  bool deep = lazy && deep_frac > 0.f;
  int Hq[3], Wq[3], Aq = 0;
  for (int l = 0; l < 3; ++l) {
    Hq[l] = level_dim(p.out_h, 3 + l);
    Wq[l] = level_dim(p.out_w, 3 + l);
    Aq += Hq[l] * Wq[l];
  }
  if (deep) {
    if (!y->pass_host) {
      EIOKU_HIP_CHECK(hipHostMalloc((void**)&y->pass_host, 4 * sizeof(int32_t), hipHostMallocDefault));
      EIOKU_HIP_CHECK(hipEventCreateWithFlags(&y->pass_event, hipEventDisableTiming));
      y->pass_host[0] = y->pass_host[1] = y->pass_host[2] = y->pass_host[3] = 0;  // no history yet -> dense
    }
    // The pinned word is the target of the previous call's asynchronous copy: consume it only once the event behind
    // that copy has fired (no copy can be in flight then: the next one is issued further down), otherwise keep
    // deciding from the last history that did complete.  A 4-int read racing the copy could tear.
    if (y->pass_pending && hipEventQuery(y->pass_event) == hipSuccess) {
      y->pass_seen[0] = (long long)y->pass_host[0] + y->pass_host[1] + y->pass_host[2];
      y->pass_seen[1] = y->pass_host[3];
      y->pass_pending = false;
    }
    const long long passed = y->pass_seen[0], total = y->pass_seen[1];
    deep = total > 0 && (float)passed <= deep_frac * (float)total;
    int32_t* f = y->flat;
    int32_t* fcnt = f + (size_t)10 * n * Aq;
    size_t off1 = 0, off0 = (size_t)n * Aq;
    for (int l = 0; l < 3 && deep; ++l) {
      const int i0 = y->deep_idx[l][0], i1 = y->deep_idx[l][1];
      if (i0 < 0 || i1 < 0) {
        deep = false;
        break;
      }
      const Op& o0 = y->ops[i0];
      const Op& o1 = y->ops[i1];
      const ConvWeights& w0 = y->weights[o0.conv];
      const ConvWeights& w1 = y->weights[o1.conv];
      auto ok3 = [](const ConvWeights& w, const Op& o) {
        return w.ks == 3 && w.stride == 1 && w.cout == 64 && w.cin % 32 == 0 && o.act == kActSiLU && o.res_buf < 0 && o.f32_out < 0;
      };
      if (!ok3(w0, o0) || !ok3(w1, o1) || (long long)n * Hq[l] * Wq[l] >= (1ll << 31)) {
        deep = false;
        break;
      }
      lbx.c0[l] = LazyConv3{y->bufs[o0.in_buf].ptr + o0.in_off, y->bufs[o0.in_buf].ch, w0.nchunks,
                            reinterpret_cast<const uint4*>(w0.d_w), 16 * w0.nf, w0.d_b,
                            y->bufs[o0.out_buf].ptr + o0.out_off, y->bufs[o0.out_buf].ch};
      lbx.c1[l] = LazyConv3{y->bufs[o1.in_buf].ptr + o1.in_off, y->bufs[o1.in_buf].ch, w1.nchunks,
                            reinterpret_cast<const uint4*>(w1.d_w), 16 * w1.nf, w1.d_b,
                            y->bufs[o1.out_buf].ptr + o1.out_off, y->bufs[o1.out_buf].ch};
      lbx.flat1[l] = f + off1;
      lbx.flat0[l] = f + off0;
      off1 += (size_t)n * Hq[l] * Wq[l];
      off0 += (size_t)9 * n * Hq[l] * Wq[l];
    }
    lbx.fcnt = fcnt;
    lbx.deep = deep;
  }
  // the counters are needed (for the next call's decision) whenever the lazy path runs
  rc = run_network(y, n, p.out_h, p.out_w, stream, fuse ? &fi : nullptr, cm, lazy, deep);
  if (rc) return rc;

  int Hl[3], Wl[3], A = 0;
  for (int l = 0; l < 3; ++l) {
    Hl[l] = level_dim(p.out_h, 3 + l);
    Wl[l] = level_dim(p.out_w, 3 + l);
    A += Hl[l] * Wl[l];
  }
  rc = ensure(&y->dets, &y->dets_cap, (size_t)n * max_det * sizeof(Det));
  if (rc) return rc;
  EIOKU_HIP_CHECK(hipMemsetAsync(y->counts, 0, (size_t)n * 2 * sizeof(int32_t), stream));
  const float* box[3] = {y->head[0], y->head[1], y->head[2]};
  const float* cls[3] = {y->head[3], y->head[4], y->head[5]};
  const unsigned long long* cmw[3] = {y->clsmax[0], y->clsmax[1], y->clsmax[2]};
  if (lazy) {
    int32_t* lvl_counts = y->lvl + (size_t)n * A;
    EIOKU_HIP_CHECK(hipMemsetAsync(lvl_counts, 0, (size_t)n * 3 * sizeof(int32_t), stream));
    if (lbx.deep) EIOKU_HIP_CHECK(hipMemsetAsync(lbx.fcnt, 0, 6 * sizeof(int32_t), stream));
    float* boxw[3] = {y->head[0], y->head[1], y->head[2]};
    rc = decode_lazy_forward(boxw, cmw, lbx, n, Hl, Wl, y->nc, conf, y->cands, y->counts, y->lvl, lvl_counts, A, stream);
    if (rc == EIOKU_OK && y->pass_host) rc = record_pass_rate(y, n, A, stream);  // history for the next call's choice
  } else {
    rc = decode_forward(box, cm ? nullptr : cls, n, Hl, Wl, y->nc, conf, y->cands, y->counts, A, stream, cm ? cmw : nullptr);
  }
  if (rc) return rc;
  ScaleParams sp;
  sp.gain = gain;
  sp.pad_x = (float)lb_geom[7];
  sp.pad_y = (float)lb_geom[8];
  sp.src_w = (float)w;
  sp.src_h = (float)h;
  rc = nms_forward(y->cands, y->counts, n, A, iou, max_det, 7680.0f, sp, y->dets, y->counts + n, stream);
  if (rc) return rc;
  const hipMemcpyKind kind = mem == EIOKU_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  EIOKU_HIP_CHECK(hipMemcpyAsync(dets_out, y->dets, (size_t)n * max_det * sizeof(Det), kind, stream));
  EIOKU_HIP_CHECK(hipMemcpyAsync(counts_out, y->counts + n, (size_t)n * sizeof(int32_t), kind, stream));
  if (mem == EIOKU_MEM_HOST) EIOKU_HIP_CHECK(hipStreamSynchronize(stream));
  return EIOKU_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Stand-alone stage entry points (parity tests; callers that own their own network)
// ---------------------------------------------------------------------------------------------
extern "C" {

int eioku_letterbox_f16(const uint8_t* bgr_dev, int n, int h, int w, const int32_t* lb_geom, const int32_t* xofs,
                        const int32_t* yofs, const int16_t* xalpha, const int16_t* ybeta, void* out_nhwc8_dev,
                        void* stream_) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(lb_geom && (n == 0 || (bgr_dev && out_nhwc8_dev)), "NULL argument");
  hipStream_t stream = (hipStream_t)stream_;
  LetterboxPlan p;
  p.src_h = h; p.src_w = w;
  p.new_h = lb_geom[0]; p.new_w = lb_geom[1]; p.top = lb_geom[2]; p.left = lb_geom[3];
  p.out_h = lb_geom[4]; p.out_w = lb_geom[5]; p.mode = lb_geom[6];
  EIOKU_REQUIRE(p.mode >= 0 && p.mode <= 2, "bad letterbox mode %d", p.mode);
  EIOKU_REQUIRE(p.mode != 1 || (xofs && yofs && xalpha && ybeta), "bilinear letterbox needs its tables");
  char* t = (char*)scratch(kSlotWork0, (size_t)(p.new_w + p.new_h) * 8 + 64);
  if (!t) return EIOKU_ENOMEM;
  p.xofs = (const int32_t*)t;
  p.yofs = (const int32_t*)(t + (size_t)p.new_w * 4);
  p.xalpha = (const int16_t*)(t + (size_t)p.new_w * 4 + (size_t)p.new_h * 4);
  p.ybeta = (const int16_t*)(t + (size_t)p.new_w * 8 + (size_t)p.new_h * 4);
  if (p.mode == 1) {
    for (int i = 0; i < p.new_w; ++i) EIOKU_REQUIRE(xofs[i] >= 0 && xofs[i] < w, "xofs[%d]=%d outside the frame", i, xofs[i]);
    EIOKU_HIP_CHECK(hipMemcpyAsync((void*)p.xofs, xofs, (size_t)p.new_w * 4, hipMemcpyHostToDevice, stream));
    EIOKU_HIP_CHECK(hipMemcpyAsync((void*)p.yofs, yofs, (size_t)p.new_h * 4, hipMemcpyHostToDevice, stream));
    EIOKU_HIP_CHECK(hipMemcpyAsync((void*)p.xalpha, xalpha, (size_t)p.new_w * 4, hipMemcpyHostToDevice, stream));
    EIOKU_HIP_CHECK(hipMemcpyAsync((void*)p.ybeta, ybeta, (size_t)p.new_h * 4, hipMemcpyHostToDevice, stream));
  }
  return letterbox_forward(bgr_dev, n, p, (__half*)out_nhwc8_dev, stream);
}

int eioku_yolo_postprocess(const float* const* box_dev, const float* const* cls_dev, int n, const int* hl,
                           const int* wl, int nc, float conf, float iou, int max_det, float gain, int pad_x,
                           int pad_y, int src_w, int src_h, void* dets_dev, int32_t* counts_dev, void* stream_) {
  EIOKU_REQUIRE_INIT();
  EIOKU_REQUIRE(box_dev && cls_dev && hl && wl && dets_dev && counts_dev, "NULL argument");
  hipStream_t stream = (hipStream_t)stream_;
  if (n == 0) return EIOKU_OK;
  int A = 0;
  for (int l = 0; l < 3; ++l) A += hl[l] * wl[l];
  Cand* cands = (Cand*)scratch(kSlotWork1, (size_t)n * A * (sizeof(Cand) + 8));
  int32_t* counts = (int32_t*)scratch(kSlotWork2, (size_t)n * 4);
  if (!cands || !counts) return EIOKU_ENOMEM;
  EIOKU_HIP_CHECK(hipMemsetAsync(counts, 0, (size_t)n * 4, stream));
  const float* box[3] = {box_dev[0], box_dev[1], box_dev[2]};
  const float* cls[3] = {cls_dev[0], cls_dev[1], cls_dev[2]};
  int rc = decode_forward(box, cls, n, hl, wl, nc, conf, cands, counts, A, stream);
  if (rc) return rc;
  ScaleParams sp{gain, (float)pad_x, (float)pad_y, (float)src_w, (float)src_h};
  return nms_forward(cands, counts, n, A, iou, max_det, 7680.0f, sp, (Det*)dets_dev, counts_dev, stream);
}

}  // extern "C"
