// IqData's fp32 shadow (blah2_amd/host/data/IqData.h: attach_shadow) against a model of what the device context does
// with it: every reported stretch of ring positions is copied from the shadow into a "device ring"; whenever
// shadow_valid() holds, the FIFO's samples must be found there at their ring positions.  GPU-free.
#include "data/IqData.h"

#include <cstdio>
#include <random>
#include <vector>

static int failures = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); failures++; } } while (0)

struct Model {
  std::vector<float> shadow, dev;
  size_t cap;
  int hooks = 0;
  explicit Model(size_t n) : shadow(2 * n, -1.f), dev(2 * n, -2.f), cap(n) {}
  void flush(IqData *q)
  {
    size_t start = 0, cnt = 0;
    q->shadow_take_pending(start, cnt);
    CHECK(cnt <= cap + 1);
    for (size_t i = 0; i < cnt; i++) {
      const size_t p = (start + i) % cap;
      dev[2 * p] = shadow[2 * p];
      dev[2 * p + 1] = shadow[2 * p + 1];
    }
  }
  static void hook(IqData *q, void *u) { static_cast<Model *>(u)->hooks++; static_cast<Model *>(u)->flush(q); }
  // the FIFO's samples as the device would read them
  bool matches(IqData *q)
  {
    flush(q);
    const auto d = q->get_data();
    for (size_t i = 0; i < d.size(); i++) {
      const size_t p = (q->head_pos() + i) % q->ring_capacity();
      if (dev[2 * p] != (float)d[i].real() || dev[2 * p + 1] != (float)d[i].imag()) return false;
    }
    return true;
  }
};

int main()
{
  const uint32_t n = 1000;
  std::mt19937 gen(3);
  std::uniform_int_distribution<int> u(-30000, 30000);
  auto sample = [&] { return std::complex<double>(u(gen), u(gen)); };
  {
    IqData q{n};
    Model m(n);
    for (int i = 0; i < 300; i++) q.push_back(sample());
    CHECK(q.attach_shadow(m.shadow.data(), 64, &Model::hook, &m));
    CHECK(q.ring_capacity() == n && q.head_pos() == 0 && q.get_length() == 300);
    CHECK(!q.shadow_valid());                    // 300 samples without a shadow
    for (int i = 0; i < 500; i++) q.push_back(sample());
    CHECK(!q.shadow_valid());
    q.drop_front(299);
    CHECK(!q.shadow_valid());                    // one old sample left
    q.drop_front(1);
    CHECK(q.shadow_valid() && m.matches(&q));    // 500 mirrored samples
    CHECK(m.hooks == 500 / 64);
    // fill up and run over: evictions move the front, positions wrap
    for (int i = 0; i < 1700; i++) q.push_back(sample());
    CHECK(q.get_length() == n && q.shadow_valid() && m.matches(&q));
    CHECK(q.head_pos() == (300 + 500 + 1700) % n);
    // one CPI's worth consumed, leftovers stay, the next CPI pushes over them (blah2.cpp's x, y)
    for (int c = 0; c < 3; c++) {
      q.drop_front(950);
      CHECK(q.get_length() == 50 && q.shadow_valid());
      for (uint32_t i = 0; i < n; i++) q.push_back(sample());
      CHECK(q.get_length() == n && q.shadow_valid() && m.matches(&q));
    }
    // the most recent samples removed (WienerHopf keeps the first nSamples)
    q.keep_front(700);
    CHECK(q.get_length() == 700 && q.shadow_valid() && m.matches(&q));
    for (int i = 0; i < 100; i++) q.push_back(sample());
    CHECK(q.shadow_valid() && m.matches(&q));
    // pops one by one
    for (int i = 0; i < 10; i++) (void)q.pop_front();
    CHECK(q.shadow_valid() && m.matches(&q));
    q.clear();
    CHECK(q.shadow_valid() && q.get_length() == 0);
    for (int i = 0; i < 70; i++) q.push_back(sample());
    CHECK(q.shadow_valid() && m.matches(&q));
    q.detach_shadow();
    CHECK(!q.shadow_valid());
    q.push_back(sample()); // no writes through the detached pointer
  }
  {
    IqData unbounded{0};
    std::vector<float> buf(16);
    CHECK(!unbounded.attach_shadow(buf.data(), 4, &Model::hook, nullptr));
  }
  // device-only front written back into the ring: those samples have no shadow until they have left
  {
    struct Src : IqDeviceFront {
      void read(uint32_t first, uint32_t count, std::complex<double> *dst) override
      {
        for (uint32_t i = 0; i < count; i++) dst[i] = {1000.0 + first + i, -5.0};
      }
    } src;
    IqData q{n};
    Model m(n);
    CHECK(q.attach_shadow(m.shadow.data(), 128, &Model::hook, &m));
    for (uint32_t i = 0; i < n; i++) q.push_back(sample());
    CHECK(q.shadow_valid() && m.matches(&q));
    q.set_device_front(n, &src);
    CHECK(q.shadow_valid());                     // still describes what was pushed; the classes use the device front instead
    const auto d = q.get_data();                 // materialises
    CHECK(d[7] == std::complex<double>(1007.0, -5.0));
    CHECK(!q.shadow_valid());
    q.drop_front(n - 20);
    for (uint32_t i = 0; i < n - 20; i++) q.push_back(sample());
    CHECK(!q.shadow_valid());                    // 20 written-back samples are still in front
    q.drop_front(20);
    CHECK(q.shadow_valid() && m.matches(&q));
  }
  std::printf(failures ? "FAILED (%d)\n" : "OK\n", failures);
  return failures ? 1 : 0;
}
