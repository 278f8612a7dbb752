/* TEST INFRASTRUCTURE ONLY -- part of oracle/ (see oracle/README.md).
 *
 * Minimal stand-in for the FFTW3 C API so that the reference's own sources
 * (/root/reference/src/process/ambiguity/Ambiguity.cpp:73-80,120-129,160 and
 *  /root/reference/src/process/clutter/WienerHopf.cpp:31-44,72-153) compile
 * unmodified in an image that has no libfftw3.  Only the entry points those
 * files call are declared.  FFTW 3.3.x (Ubuntu 22.04 libfftw3-dev, see the
 * reference Dockerfile:12) computes an unnormalised DFT
 *     out[k] = sum_n in[n] * exp(sign * 2*pi*i * n*k / N)
 * which is a fully specified mathematical object; oracle/shim/fft64.cpp
 * evaluates the same sum in fp64.
 */
#ifndef BLAH2_ORACLE_FFTW3_SHIM_H
#define BLAH2_ORACLE_FFTW3_SHIM_H

#ifdef __cplusplus
extern "C" {
#endif

typedef double fftw_complex[2];
typedef struct oracle_fft_plan_s *fftw_plan;

#define FFTW_FORWARD (-1)
#define FFTW_BACKWARD (+1)
#define FFTW_MEASURE (0U)
#define FFTW_ESTIMATE (1U << 6)

fftw_plan fftw_plan_dft_1d(int n, fftw_complex *in, fftw_complex *out,
                           int sign, unsigned flags);
void fftw_execute(const fftw_plan p);
void fftw_destroy_plan(fftw_plan p);
int fftw_init_threads(void);
void fftw_plan_with_nthreads(int nthreads);
void fftw_cleanup_threads(void);

#ifdef __cplusplus
}
#endif
#endif
