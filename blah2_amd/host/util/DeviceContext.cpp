#include "util/DeviceContext.h"

#include "blah2hip.h"
#include "process/ambiguity/Ambiguity.h"

#include <algorithm>
#include <cstdlib>
#include <stdexcept>
#include <string>

namespace {
void chk(int rc, const char *what)
{
  if (rc != BLAH2HIP_OK) throw std::runtime_error(std::string(what) + ": " + blah2hip_last_error());
}
constexpr unsigned N_WORKERS = 8; // narrowing 2 M samples: 0.25 ms with eight threads, 1.7 ms with one
constexpr size_t SHADOW_CHUNK = 1u << 18;       // samples per eager upload (2 MB)
constexpr size_t SHADOW_MAX = 64u << 20;        // FIFOs of more samples (a capture buffer of minutes) stay on the per-CPI path
} // namespace

DeviceContext &DeviceContext::get()
{
  static DeviceContext *inst = new DeviceContext; // never destroyed: IqData objects with static storage may outlive main()
  return *inst;
}

DeviceContext::DeviceContext()
{
  chk(blah2hip_ctx_create(Ambiguity::default_device(), &ctx), "DeviceContext");
  IqData::destroyed_hook = &DeviceContext::forget;
  for (unsigned i = 0; i < N_WORKERS; i++) workers.emplace_back([this] { worker(); });
}

DeviceContext::~DeviceContext() {}

void *DeviceContext::stream() const { return blah2hip_ctx_stream(ctx); }
void DeviceContext::sync()
{
  chk(blah2hip_ctx_sync(ctx), "DeviceContext::sync");
  uploadsSinceSync = 0;
}
void DeviceContext::d2h(void *hptr, const void *dptr, size_t bytes) { chk(blah2hip_ctx_d2h(ctx, hptr, dptr, bytes), "DeviceContext::d2h"); }

void *DeviceContext::alloc_device(size_t bytes)
{
  void *p = nullptr;
  chk(blah2hip_ctx_malloc(ctx, bytes, &p), "DeviceContext::alloc_device");
  return p;
}
void *DeviceContext::alloc_pinned(size_t bytes)
{
  void *p = nullptr;
  chk(blah2hip_ctx_malloc_host(ctx, bytes, &p), "DeviceContext::alloc_pinned");
  return p;
}
void DeviceContext::free_device(void *p) { if (p) { (void)blah2hip_ctx_sync(ctx); (void)blah2hip_ctx_free(ctx, p); } }
void DeviceContext::free_pinned(void *p) { if (p) { (void)blah2hip_ctx_sync(ctx); (void)blah2hip_ctx_free_host(ctx, p); } }

void DeviceContext::forget(IqData *q)
{
  DeviceContext &c = get();
  std::lock_guard<std::recursive_mutex> api_(c.api);
  std::lock_guard<std::mutex> lk(c.mu);
  auto it = c.mirrors.find(q);
  if (it == c.mirrors.end()) return;
  Mirror *m = it->second;
  (void)blah2hip_ctx_sync(c.ctx);
  (void)blah2hip_ctx_free(c.ctx, m->dev);
  (void)blah2hip_ctx_free(c.ctx, m->devFront);
  (void)blah2hip_ctx_free(c.ctx, m->devRing);
  if (m->shadow) (void)blah2hip_ctx_free_host(c.ctx, m->shadow);
  delete m;
  c.mirrors.erase(it);
}

DeviceContext::Mirror &DeviceContext::mirror_of(IqData *q)
{
  std::lock_guard<std::mutex> lk(mu);
  Mirror *&m = mirrors[q];
  if (!m) { m = new Mirror; m->ctx = this; }
  return *m;
}

// The staging buffer for the next upload.  The buffer handed out two uploads ago is reused: the copy that read it has
// been followed by at least one sync() of a class waiting for its results in every sequence the classes run (Spectrum,
// WienerHopf and Ambiguity each end with one), and when in doubt -- a reallocation, or three uploads without a
// sync in between -- the stream is drained first.
float *DeviceContext::staging(size_t samples)
{
  const int k = pinnedNext;
  pinnedNext ^= 1;
  if (uploadsSinceSync >= 2) sync();
  if (pinnedSamples[k] < samples) {
    sync();
    if (pinned[k]) chk(blah2hip_ctx_free_host(ctx, pinned[k]), "DeviceContext: free pinned");
    pinned[k] = nullptr;
    pinnedSamples[k] = 0;
    void *p = nullptr;
    chk(blah2hip_ctx_malloc_host(ctx, samples * 2 * sizeof(float), &p), "DeviceContext: pinned staging");
    pinned[k] = (float *)p;
    pinnedSamples[k] = samples;
  }
  uploadsSinceSync++;
  return pinned[k];
}

// ---- the eager path ---------------------------------------------------------------------------------
void DeviceContext::on_chunk(IqData *q, void *mirror)
{
  Mirror &m = *static_cast<Mirror *>(mirror);
  m.ctx->flush_pending(q, m);
}

// the stretch of ring positions pushed since the last call goes to the same positions of the device ring
void DeviceContext::flush_pending(IqData *q, Mirror &m)
{
  std::lock_guard<std::recursive_mutex> api_(api);
  size_t start = 0, cnt = 0;
  q->shadow_take_pending(start, cnt);
  while (cnt) {
    const size_t a = std::min(cnt, m.ringCap - start);
    chk(blah2hip_ctx_h2d(ctx, m.devRing + 2 * start, m.shadow + 2 * start, a * 2 * sizeof(float)), "DeviceContext: eager upload");
    cnt -= a;
    start = 0;
  }
}

void DeviceContext::try_attach(IqData *q, Mirror &m)
{
  m.tried = true;
  const size_t n = q->get_n();
  static const bool off = std::getenv("BLAH2HIP_NO_EAGER_UPLOAD") != nullptr; // measurements: the per-CPI path only
  if (off || !n || n > SHADOW_MAX) return;
  // best effort: the CPI at hand has already been served by the per-CPI path, so a failed allocation (64 M samples are
  // 512 MB of HBM and of pinned memory) leaves this FIFO on that path instead of failing the call
  void *h = nullptr, *d = nullptr;
  if (blah2hip_ctx_malloc_host(ctx, n * 2 * sizeof(float), &h) != 0 || !h) return;
  if (blah2hip_ctx_malloc(ctx, n * 2 * sizeof(float), &d) != 0 || !d) {
    (void)blah2hip_ctx_free_host(ctx, h);
    return;
  }
  m.shadow = (float *)h;
  m.devRing = (float *)d;
  m.ringCap = n;
  if (!q->attach_shadow(m.shadow, SHADOW_CHUNK, &DeviceContext::on_chunk, &m)) {
    (void)blah2hip_ctx_free_host(ctx, h);
    (void)blah2hip_ctx_free(ctx, d);
    m.shadow = m.devRing = nullptr;
    m.ringCap = 0;
  }
}

const void *DeviceContext::resident(IqData *q, uint32_t count)
{
  std::lock_guard<std::recursive_mutex> api_(api);
  Mirror &m = mirror_of(q);
  if (count > q->get_length()) throw std::runtime_error("Attempting to pop from an empty deque"); // what the reference's pops would throw
  if (m.gen == q->generation() && m.view && m.viewCount >= count) return m.view;
  if (m.ringCap && q->shadow_valid() && !q->device_front_count()) {
    // everything the FIFO holds was narrowed as it was pushed and all but the last stretch is on its way (or there)
    flush_pending(q, m);
    const size_t head = q->head_pos(), all = q->get_length();
    const size_t lin = std::min(all, m.ringCap - head);
    if (count <= lin) {
      m.view = m.devRing + 2 * head;
      m.viewCount = (uint32_t)lin;
    } else { // the front wraps around the end of the ring: two device copies make it one plane
      if (m.cap < all) {
        sync();
        if (m.dev) chk(blah2hip_ctx_free(ctx, m.dev), "DeviceContext: free");
        m.dev = nullptr;
        m.cap = 0;
        void *d = nullptr;
        chk(blah2hip_ctx_malloc(ctx, all * 2 * sizeof(float), &d), "DeviceContext: device plane");
        m.dev = (float *)d;
        m.cap = all;
      }
      chk(blah2hip_ctx_d2d(ctx, m.dev, m.devRing + 2 * head, lin * 2 * sizeof(float)), "DeviceContext: gather");
      chk(blah2hip_ctx_d2d(ctx, m.dev + 2 * lin, m.devRing, (all - lin) * 2 * sizeof(float)), "DeviceContext: gather");
      m.view = m.dev;
      m.viewCount = (uint32_t)all;
    }
    m.gen = q->generation();
    return m.view;
  }
  // upload everything the FIFO holds: the next class asks for a little more or less of the same CPI
  const uint32_t all = q->get_length();
  if (q->device_front_count()) (void)q->get_data(); // device-only samples under a view we lost track of: bring them home first (rare)
  if (m.cap < all) {
    sync();
    if (m.dev) chk(blah2hip_ctx_free(ctx, m.dev), "DeviceContext: free");
    m.dev = nullptr;
    m.cap = 0;
    void *d = nullptr;
    chk(blah2hip_ctx_malloc(ctx, (size_t)all * 2 * sizeof(float), &d), "DeviceContext: device plane");
    m.dev = (float *)d;
    m.cap = all;
  }
  // (Sending the channel in four pieces, so that the DMA of one runs while the next is narrowed, shaved 0.07 ms off the
  // upload and added 2.1 ms to the NEXT stage's device-to-host copy on every box tried; one copy per channel it is.)
  float *dst = staging(all);
  parallel_for(all, 1 << 16, [q, dst](size_t a, size_t b) { q->copy_front_c32((uint32_t)a, (uint32_t)(b - a), dst + 2 * a); });
  chk(blah2hip_ctx_h2d(ctx, m.dev, dst, (size_t)all * 2 * sizeof(float)), "DeviceContext: upload");
  m.view = m.dev;
  m.viewCount = all;
  m.gen = q->generation();
  if (!m.tried) try_attach(q, m); // from the next CPI on the samples arrive narrowed and uploaded
  return m.view;
}

void *DeviceContext::front_buffer(IqData *q, uint32_t count)
{
  std::lock_guard<std::recursive_mutex> api_(api);
  Mirror &m = mirror_of(q);
  if (m.capFront < count) {
    sync();
    if (m.devFront) chk(blah2hip_ctx_free(ctx, m.devFront), "DeviceContext: free");
    m.devFront = nullptr;
    m.capFront = 0;
    void *d = nullptr;
    chk(blah2hip_ctx_malloc(ctx, (size_t)count * 2 * sizeof(float), &d), "DeviceContext: device plane");
    m.devFront = (float *)d;
    m.capFront = count;
  }
  return m.devFront;
}

void DeviceContext::adopt_front(IqData *q, uint32_t count)
{
  std::lock_guard<std::recursive_mutex> api_(api);
  Mirror &m = mirror_of(q);
  q->set_device_front(count, &m); // bumps the generation
  m.view = m.devFront;
  m.viewCount = std::min<uint32_t>(count, q->get_length());
  m.gen = q->generation();
}

void DeviceContext::consumed(IqData *q, uint32_t count)
{
  std::lock_guard<std::recursive_mutex> api_(api);
  Mirror &m = mirror_of(q);
  if (!m.view) return;
  const uint32_t d = std::min(count, m.viewCount);
  m.view += 2 * (size_t)d;
  m.viewCount -= d;
  m.gen = q->generation(); // the caller has just dropped exactly these samples
}

// device-only front samples -> host doubles (IqData::materialise)
void DeviceContext::Mirror::read(uint32_t first, uint32_t count, std::complex<double> *dst)
{
  std::lock_guard<std::recursive_mutex> api_(ctx->api);
  std::vector<float> tmp(2 * (size_t)count);
  ctx->d2h(tmp.data(), devFront + 2 * (size_t)first, tmp.size() * sizeof(float));
  ctx->sync();
  for (uint32_t i = 0; i < count; i++) dst[i] = {tmp[2 * i], tmp[2 * i + 1]};
}

// ---- a few worker threads for the narrowing / widening loops -------------------------------------
void DeviceContext::worker()
{
  uint64_t seen = 0;
  for (;;) {
    std::unique_lock<std::mutex> lk(pmu);
    pcv.wait(lk, [&] { return stopping || (job && jobId != seen && jobNext < jobN); });
    if (stopping) return;
    const uint64_t id = jobId;
    while (job && jobId == id && jobNext < jobN) {
      const size_t a = jobNext, b = std::min(jobN, a + jobGrain);
      jobNext = b;
      const auto *fn = job;
      lk.unlock();
      (*fn)(a, b);
      lk.lock();
      if (--jobPending == 0) pdone.notify_all();
    }
    seen = id;
  }
}

void DeviceContext::parallel_for(size_t n, size_t grain, const std::function<void(size_t, size_t)> &fn)
{
  std::lock_guard<std::recursive_mutex> api_(api);
  if (n <= grain || workers.empty()) {
    if (n) fn(0, n);
    return;
  }
  std::unique_lock<std::mutex> lk(pmu);
  job = &fn;
  jobN = n;
  jobGrain = grain;
  jobNext = 0;
  jobPending = (n + grain - 1) / grain;
  jobId++;
  pcv.notify_all();
  // the calling thread takes chunks too
  const uint64_t id = jobId;
  while (jobId == id && jobNext < jobN) {
    const size_t a = jobNext, b = std::min(jobN, a + jobGrain);
    jobNext = b;
    lk.unlock();
    fn(a, b);
    lk.lock();
    if (--jobPending == 0) pdone.notify_all();
  }
  pdone.wait(lk, [&] { return jobPending == 0; });
  job = nullptr;
}
