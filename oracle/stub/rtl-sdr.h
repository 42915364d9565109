/* Stub of librtlsdr's public header, written for this repo (test infrastructure).
 *
 * The upstream dump1090.c includes "rtl-sdr.h" (dump1090.c:46) and references
 * 14 librtlsdr entry points (dump1090.c:151,390-433,520,3008).  librtlsdr is
 * not vendored upstream and is absent from this image.  With --ifile none of
 * these is reached except rtlsdr_close(NULL) at dump1090.c:3008, so inert
 * inline definitions are enough to build the *unmodified* reference sources
 * into oracle/_ref/ (see oracle/Makefile).  Nothing here is product code. */
#ifndef ORACLE_STUB_RTL_SDR_H
#define ORACLE_STUB_RTL_SDR_H
#include <stdint.h>

typedef struct rtlsdr_dev rtlsdr_dev_t;
typedef void (*rtlsdr_read_async_cb_t)(unsigned char *buf, uint32_t len, void *ctx);

static inline uint32_t rtlsdr_get_device_count(void) { return 0; }
static inline int rtlsdr_get_device_usb_strings(uint32_t i, char *m, char *p, char *s)
{ (void)i; if (m) m[0] = 0; if (p) p[0] = 0; if (s) s[0] = 0; return -1; }
static inline int rtlsdr_open(rtlsdr_dev_t **d, uint32_t i) { (void)d; (void)i; return -1; }
static inline int rtlsdr_close(rtlsdr_dev_t *d) { (void)d; return 0; }
static inline int rtlsdr_set_tuner_gain_mode(rtlsdr_dev_t *d, int m) { (void)d; (void)m; return -1; }
static inline int rtlsdr_get_tuner_gains(rtlsdr_dev_t *d, int *g) { (void)d; (void)g; return 0; }
static inline int rtlsdr_set_tuner_gain(rtlsdr_dev_t *d, int g) { (void)d; (void)g; return -1; }
static inline int rtlsdr_get_tuner_gain(rtlsdr_dev_t *d) { (void)d; return 0; }
static inline int rtlsdr_set_freq_correction(rtlsdr_dev_t *d, int p) { (void)d; (void)p; return -1; }
static inline int rtlsdr_set_agc_mode(rtlsdr_dev_t *d, int on) { (void)d; (void)on; return -1; }
static inline int rtlsdr_set_center_freq(rtlsdr_dev_t *d, uint32_t f) { (void)d; (void)f; return -1; }
static inline int rtlsdr_set_sample_rate(rtlsdr_dev_t *d, uint32_t r) { (void)d; (void)r; return -1; }
static inline int rtlsdr_reset_buffer(rtlsdr_dev_t *d) { (void)d; return -1; }
static inline int rtlsdr_read_async(rtlsdr_dev_t *d, rtlsdr_read_async_cb_t cb, void *ctx,
                                    uint32_t n, uint32_t len)
{ (void)d; (void)cb; (void)ctx; (void)n; (void)len; return -1; }
#endif
