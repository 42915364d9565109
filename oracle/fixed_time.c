/* LD_PRELOAD interposer: time() returns a constant (test infrastructure only).
 *
 * The reference's ICAO whitelist expires entries on wall-clock
 * (dump1090.c:913,924: TTL 60 s on time(NULL)).  A CPU run over a multi-GiB
 * stream takes longer than 60 s, so AP-validated messages (DF0/4/5/16/20/21)
 * would depend on how fast the host is.  Running oracle/_ref/dump1090_ref
 * under this shim makes the reference a pure function of its input bytes;
 * the product host uses the same "TTL never expires inside one file run"
 * semantics.  SURVEY.md section 7 hard part 3. */
/* gettimeofday() is pinned too: the reference's aircraft tracker (only active with an SBS or HTTP
 * client, dump1090.c:1806) stamps CPR frames with mstime() (dump1090.c:287, 2114-2118) and picks the
 * newer of an even/odd pair (dump1090.c:1973); with a constant clock that choice - hence every
 * position on the SBS port - is a function of the input alone. */
#include <stddef.h>
#include <sys/time.h>
#include <time.h>
time_t time(time_t *t) {
    const time_t fixed = (time_t)1700000000;
    if (t) *t = fixed;
    return fixed;
}
int gettimeofday(struct timeval *tv, void *tz) {
    (void)tz;
    if (tv) { tv->tv_sec = (time_t)1700000000; tv->tv_usec = 0; }
    return 0;
}
