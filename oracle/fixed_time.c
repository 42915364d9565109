/* LD_PRELOAD interposer: time() returns a constant (test infrastructure only).
 *
 * The reference's ICAO whitelist expires entries on wall-clock
 * (dump1090.c:913,924: TTL 60 s on time(NULL)).  A CPU run over a multi-GiB
 * stream takes longer than 60 s, so AP-validated messages (DF0/4/5/16/20/21)
 * would depend on how fast the host is.  Running oracle/_ref/dump1090_ref
 * under this shim makes the reference a pure function of its input bytes;
 * the product host uses the same "TTL never expires inside one file run"
 * semantics.  SURVEY.md section 7 hard part 3. */
#include <time.h>
time_t time(time_t *t) {
    const time_t fixed = (time_t)1700000000;
    if (t) *t = fixed;
    return fixed;
}
