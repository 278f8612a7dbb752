// Ambiguity: delay-Doppler cross-ambiguity map of one CPI, computed on an
// MI355X through the C ABI of include/blah2hip.h.  Constructor arguments,
// getters, ownership and error behaviour follow the reference class
// (src/process/ambiguity/Ambiguity.h:34-58, Ambiguity.cpp:11-199):
//   * process() consumes nCorr*nDopplerBins samples from both FIFOs by
//     pop_front (so an underfull FIFO throws std::runtime_error),
//   * the returned Map is owned by this object and overwritten on every call,
//   * not re-entrant: one thread per object.
#ifndef BLAH2HIP_HOST_AMBIGUITY_H
#define BLAH2HIP_HOST_AMBIGUITY_H

#include "data/IqData.h"
#include "data/Map.h"
#include "process/meta/HammingNumber.h"

#include <memory>
#include <stdint.h>
#include <vector>

class Ambiguity
{
public:
  using Complex = std::complex<double>;

  Ambiguity(int32_t delayMin, int32_t delayMax, int32_t dopplerMin, int32_t dopplerMax, uint32_t fs,
            uint32_t n, bool roundHamming = false);
  ~Ambiguity();
  Ambiguity(const Ambiguity &) = delete;
  Ambiguity &operator=(const Ambiguity &) = delete;

  Map<Complex> *process(IqData *x, IqData *y);

  double get_doppler_middle() const;
  uint16_t get_n_delay_bins() const;
  uint16_t get_n_doppler_bins() const;
  uint16_t get_n_corr() const;
  double get_cpi() const;
  uint32_t get_nfft() const;
  uint32_t get_n_samples() const;

  // extension: which GPU to use (default 0, or env BLAH2HIP_DEVICE)
  static int default_device();

private:
  blah2hip_amb_s *engine = nullptr;
  uint32_t fs;
  uint32_t nSamples;
  uint16_t nDelayBins, nDopplerBins, nCorr;
  double dopplerMiddle, cpi;
  uint32_t nfft;
  std::unique_ptr<Map<Complex>> map;
  float *mapF = nullptr;   // pinned: complex fp32 map as the device wrote it
  double *metF = nullptr;  // pinned: noisePower, maxPower
};

#endif
