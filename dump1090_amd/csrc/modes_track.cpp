// Aircraft table + CPR position decode + the BaseStation (SBS, port 30003) line of a message: the
// sink-side state the reference keeps behind useModesMessage() when an SBS or HTTP client is
// connected (dump1090.c:1806-1808).  Host code, sequential like the reference's; no GPU part.
//
//   interactiveReceiveData   dump1090.c:2069-2167   -> modes_tracker_receive
//   decodeCPR                dump1090.c:1952-1990   -> cpr_airborne
//   decodeCPRSurface         dump1090.c:2004-2052   -> cpr_surface
//   decodeMovementField      dump1090.c:2056-2066   -> movement_knots
//   cprNLFunction            dump1090.c:1868-1929   -> cpr_nl (table from 1090-WP-9-14)
//   interactiveRemoveStaleAircrafts :2203-2224      -> modes_tracker_expire
//   modesSendSBSOutput       dump1090.c:2397-2448   -> modes_format_sbs
//   aircraftsToJson          dump1090.c:2505-2552   -> modes_tracker_json
//
// The reference keeps a linked list searched linearly and two process globals for the receiver's
// reference position; here: a hash index over a stable array, everything inside the tracker object.
// Time is an argument (milliseconds): the caller passes its clock, tests pass a constant - the
// reference's own dependence on wall-clock (which CPR frame is "newer", the 10 s pairing window,
// the 60 s TTL) becomes reproducible.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <new>
#include <unordered_map>

#include "modes_host.h"

namespace {

// Number of longitude zones at a latitude: upper bounds of the bands NL = 59 .. 2 (1090-WP-9-14, the
// table the reference spells out as an if-chain).  NL(lat) = 59 - (number of bounds <= |lat|).
const double kNlBound[58] = {
    10.47047130, 14.82817437, 18.18626357, 21.02939493, 23.54504487, 25.82924707, 27.93898710, 29.91135686,
    31.77209708, 33.53993436, 35.22899598, 36.85025108, 38.41241892, 39.92256684, 41.38651832, 42.80914012,
    44.19454951, 45.54626723, 46.86733252, 48.16039128, 49.42776439, 50.67150166, 51.89342469, 53.09516153,
    54.27817472, 55.44378444, 56.59318756, 57.72747354, 58.84763776, 59.95459277, 61.04917774, 62.13216659,
    63.20427479, 64.26616523, 65.31845310, 66.36171008, 67.39646774, 68.42322022, 69.44242631, 70.45451075,
    71.45986473, 72.45884545, 73.45177442, 74.43893416, 75.42056257, 76.39684391, 77.36789461, 78.33374083,
    79.29428225, 80.24923213, 81.19801349, 82.13956981, 83.07199445, 83.99173563, 84.89166191, 85.75541621,
    86.53536998, 87.00000000};

int cpr_nl(double lat) {
    if (lat < 0) lat = -lat;
    int lo = 0, hi = 58;                       // first bound that exceeds lat
    while (lo < hi) {
        const int mid = (lo + hi) / 2;
        if (lat < kNlBound[mid]) hi = mid; else lo = mid + 1;
    }
    return 59 - lo;
}
int cpr_n(double lat, int isodd) {
    const int nl = cpr_nl(lat) - isodd;
    return nl < 1 ? 1 : nl;
}
int pos_mod(int a, int b) {
    const int r = a % b;
    return r < 0 ? r + b : r;
}

}  // namespace

struct modes_tracker {
    std::deque<modes_aircraft> list;                    // stable addresses; slot reuse through `free_slots`
    std::deque<size_t> free_slots;
    std::unordered_map<uint32_t, size_t> index;         // ICAO address -> slot
    std::deque<size_t> order;                           // slots, newest aircraft first (the reference's list order)
    double ref_lat = 0, ref_lon = 0;                    // running mean of decoded airborne positions
    int ref_count = 0;
};

namespace {

// Global airborne decode from the stored even/odd pair; leaves lat/lon alone when the two frames
// disagree on the latitude band.
void cpr_airborne(modes_aircraft *a) {
    const double dlat0 = 360.0 / 60, dlat1 = 360.0 / 59;
    const double lat0 = a->even_cprlat, lat1 = a->odd_cprlat, lon0 = a->even_cprlon, lon1 = a->odd_cprlon;
    const int j = (int)std::floor(((59 * lat0 - 60 * lat1) / 131072) + 0.5);
    double rlat0 = dlat0 * (pos_mod(j, 60) + lat0 / 131072);
    double rlat1 = dlat1 * (pos_mod(j, 59) + lat1 / 131072);
    if (rlat0 >= 270) rlat0 -= 360;
    if (rlat1 >= 270) rlat1 -= 360;
    if (cpr_nl(rlat0) != cpr_nl(rlat1)) return;
    const bool use_even = a->even_cprtime > a->odd_cprtime;
    const double rlat = use_even ? rlat0 : rlat1;
    const int nl = cpr_nl(rlat);
    const int ni = cpr_n(rlat, use_even ? 0 : 1);
    const int m = (int)std::floor((((lon0 * (nl - 1)) - (lon1 * nl)) / 131072.0) + 0.5);
    a->lon = (360.0 / ni) * (pos_mod(m, ni) + (use_even ? lon0 : lon1) / 131072);
    a->lat = rlat;
    if (a->lon > 180) a->lon -= 360;
}

// Local surface decode of one frame against the receiver's reference position.
void cpr_surface(const modes_tracker *tr, modes_aircraft *a, int fflag, int raw_lat, int raw_lon) {
    if (tr->ref_count == 0) return;
    const double dlat = fflag ? 90.0 / 59 : 90.0 / 60;
    const double ref_lat = tr->ref_lat, ref_lon = tr->ref_lon;
    const int j = (int)std::floor(ref_lat / dlat) +
                  (int)std::floor(0.5 + pos_mod((int)ref_lat, (int)dlat) / dlat - (double)raw_lat / 131072);
    double lat = dlat * (j + (double)raw_lat / 131072);
    if (std::fabs(lat - ref_lat) > 45) lat += lat > ref_lat ? -90 : 90;
    if (lat < -90 || lat > 90) return;
    int ni = cpr_n(lat, fflag);
    if (ni == 0) ni = 1;
    const double dlon = 90.0 / ni;
    const int m = (int)std::floor(ref_lon / dlon) +
                  (int)std::floor(0.5 + pos_mod((int)ref_lon, (int)dlon) / dlon - (double)raw_lon / 131072);
    double lon = dlon * (m + (double)raw_lon / 131072);
    while (lon > ref_lon + 45) lon -= 90;
    while (lon < ref_lon - 45) lon += 90;
    if (lon > 180) lon -= 360;
    if (lon < -180) lon += 360;
    a->lat = lat;
    a->lon = lon;
}

// Ground speed of a surface movement field, knots truncated to int like the reference's return type.
int movement_knots(int mv) {
    if (mv == 0) return -1;
    if (mv == 1) return 0;
    if (mv <= 8) return (int)((mv - 2) * 0.125 + 0.125);
    if (mv <= 12) return (int)((mv - 9) * 0.25 + 1);
    if (mv <= 38) return (int)((mv - 13) * 0.5 + 2);
    if (mv <= 93) return (mv - 39) + 15;
    if (mv <= 108) return (mv - 94) * 2 + 70;
    if (mv <= 123) return (mv - 109) * 5 + 100;
    return 175;
}

}  // namespace

extern "C" {

modes_tracker *modes_tracker_create(void) { return new (std::nothrow) modes_tracker; }
void modes_tracker_destroy(modes_tracker *tr) { delete tr; }

const modes_aircraft *modes_tracker_receive(modes_tracker *tr, const struct modesMessage *mm, int check_crc, int64_t now_ms) {
    if (!tr || !mm) return nullptr;
    if (check_crc && mm->crcok == 0) return nullptr;
    const uint32_t addr = ((uint32_t)mm->aa1 << 16) | ((uint32_t)mm->aa2 << 8) | (uint32_t)mm->aa3;
    modes_aircraft *a;
    auto it = tr->index.find(addr);
    if (it == tr->index.end()) {
        size_t slot;
        if (!tr->free_slots.empty()) { slot = tr->free_slots.front(); tr->free_slots.pop_front(); }
        else { slot = tr->list.size(); tr->list.emplace_back(); }
        a = &tr->list[slot];
        memset(a, 0, sizeof *a);
        a->addr = addr;
        snprintf(a->hexaddr, sizeof a->hexaddr, "%06x", (unsigned)addr);
        tr->index.emplace(addr, slot);
        tr->order.push_front(slot);
    } else {
        a = &tr->list[it->second];
    }
    a->seen_ms = now_ms;
    a->messages++;

    const int df = mm->msgtype, me = mm->metype;
    if (df == 0 || df == 4 || df == 20) {
        a->altitude = mm->altitude;
    } else if (df == 17 || df == 18) {
        if (me >= 1 && me <= 4) {
            memcpy(a->flight, mm->flight, sizeof a->flight);
        } else if (me >= 9 && me <= 18) {
            a->altitude = mm->altitude;
            if (mm->fflag) { a->odd_cprlat = mm->raw_latitude; a->odd_cprlon = mm->raw_longitude; a->odd_cprtime = now_ms; }
            else           { a->even_cprlat = mm->raw_latitude; a->even_cprlon = mm->raw_longitude; a->even_cprtime = now_ms; }
            if (llabs((long long)(a->even_cprtime - a->odd_cprtime)) <= 10000) {   // a fresh even/odd pair
                const double plat = a->lat, plon = a->lon;
                cpr_airborne(a);
                if (a->lat != plat || a->lon != plon) {                          // feeds the receiver's reference position
                    if (tr->ref_count == 0) { tr->ref_lat = a->lat; tr->ref_lon = a->lon; }
                    else {
                        tr->ref_lat += (a->lat - tr->ref_lat) / (tr->ref_count + 1);
                        tr->ref_lon += (a->lon - tr->ref_lon) / (tr->ref_count + 1);
                    }
                    if (tr->ref_count < 10000) tr->ref_count++;
                }
            }
        } else if (me >= 5 && me <= 8) {
            if (tr->ref_count) {
                if (mm->ground_track_valid) a->track = mm->ground_track;
                if (mm->movement_valid) a->speed = movement_knots(mm->movement);
                a->altitude = 0;
                cpr_surface(tr, a, mm->fflag, mm->raw_latitude, mm->raw_longitude);
            }
        } else if (me == 19) {
            if (mm->mesub == 1 || mm->mesub == 2) { a->speed = mm->velocity; a->track = mm->heading; }
        }
    }
    return a;
}

uint64_t modes_tracker_count(const modes_tracker *tr) { return tr ? tr->order.size() : 0; }
const modes_aircraft *modes_tracker_get(const modes_tracker *tr, uint64_t i) {
    return (tr && i < tr->order.size()) ? &tr->list[tr->order[(size_t)i]] : nullptr;
}
void modes_tracker_reference(const modes_tracker *tr, double *lat, double *lon, int *count) {
    if (lat) *lat = tr ? tr->ref_lat : 0;
    if (lon) *lon = tr ? tr->ref_lon : 0;
    if (count) *count = tr ? tr->ref_count : 0;
}

uint64_t modes_tracker_expire(modes_tracker *tr, int64_t now_ms, int64_t ttl_ms) {
    if (!tr) return 0;
    uint64_t gone = 0;
    for (auto it = tr->order.begin(); it != tr->order.end();) {
        modes_aircraft &a = tr->list[*it];
        if (now_ms - a.seen_ms > ttl_ms) {
            tr->index.erase(a.addr);
            tr->free_slots.push_back(*it);
            it = tr->order.erase(it);
            gone++;
        } else {
            ++it;
        }
    }
    return gone;
}

// aircraftsToJson (dump1090.c:2505-2552), the body of the reference's /data.json: the aircraft with a
// decoded position, newest first.  Returns the length of the text (without the terminating 0); at
// most cap - 1 characters are stored, so a return value >= cap means "call again with a larger buffer".
size_t modes_tracker_json(const modes_tracker *tr, int metric, char *buf, size_t cap) {
    size_t len = 0;
    auto put = [&](const char *text, size_t n) {
        for (size_t i = 0; i < n; i++, len++)
            if (buf && len + 1 < cap) buf[len] = text[i];
    };
    put("[\n", 2);
    bool any = false;
    if (tr) {
        for (size_t slot : tr->order) {
            const modes_aircraft &a = tr->list[slot];
            if (a.lat == 0 || a.lon == 0) continue;
            int altitude = a.altitude, speed = a.speed;
            if (metric) { altitude = (int)(altitude / 3.2828); speed = (int)(speed * 1.852); }   // the reference's constants
            char line[256];
            const int n = snprintf(line, sizeof line, "{\"hex\":\"%s\", \"flight\":\"%s\", \"lat\":%f, \"lon\":%f, \"altitude\":%d, "
                                   "\"track\":%d, \"speed\":%d},\n", a.hexaddr, a.flight, a.lat, a.lon, altitude, a.track, speed);
            put(line, (size_t)n);
            any = true;
        }
    }
    if (any) {                                           // the last ",\n" becomes "\n"
        len -= 2;
        put("\n", 1);
    }
    put("]\n", 2);
    if (buf && cap) buf[len < cap ? len : cap - 1] = 0;
    return len;
}

int modes_format_sbs(const struct modesMessage *mm, const modes_aircraft *a, char *buf, size_t cap) {
    if (!mm || !buf || cap < 2) return 0;
    int emergency = 0, ground = 0, alert = 0, spi = 0;
    const int df = mm->msgtype;
    if (df == 4 || df == 5 || df == 21) {
        if (mm->identity == 7500 || mm->identity == 7600 || mm->identity == 7700) emergency = -1;
        if (mm->fs == 1 || mm->fs == 3) ground = -1;
        if (mm->fs == 2 || mm->fs == 3 || mm->fs == 4) alert = -1;
        if (mm->fs == 4 || mm->fs == 5) spi = -1;
    }
    char icao[8];
    snprintf(icao, sizeof icao, "%02X%02X%02X", mm->aa1, mm->aa2, mm->aa3);
    const bool es = df == 17 || df == 18;
    int n;
    // field layout of a BaseStation line: MSG,type,,,ICAO,,,,,,callsign,altitude,speed,track,lat,lon,vrate,squawk,alert,emergency,spi,ground
    if (df == 0)
        n = snprintf(buf, cap, "MSG,5,,,%s,,,,,,,%d,,,,,,,,,,", icao, mm->altitude);
    else if (df == 4)
        n = snprintf(buf, cap, "MSG,5,,,%s,,,,,,,%d,,,,,,,%d,%d,%d,%d", icao, mm->altitude, alert, emergency, spi, ground);
    else if (df == 5 || df == 21)
        n = snprintf(buf, cap, "MSG,6,,,%s,,,,,,,,,,,,,%d,%d,%d,%d,%d", icao, mm->identity, alert, emergency, spi, ground);
    else if (df == 11)
        n = snprintf(buf, cap, "MSG,8,,,%s,,,,,,,,,,,,,,,,,", icao);
    else if (es && mm->metype == 4)
        n = snprintf(buf, cap, "MSG,1,,,%s,,,,,,%s,,,,,,,,0,0,0,0", icao, mm->flight);
    else if (es && mm->metype >= 9 && mm->metype <= 18) {
        if (!a) return 0;
        if (a->lat == 0 && a->lon == 0)
            n = snprintf(buf, cap, "MSG,3,,,%s,,,,,,,%d,,,,,,,0,0,0,0", icao, mm->altitude);
        else
            n = snprintf(buf, cap, "MSG,3,,,%s,,,,,,,%d,,,%1.5f,%1.5f,,,0,0,0,0", icao, mm->altitude, a->lat, a->lon);
    } else if (es && mm->metype == 19 && mm->mesub == 1) {
        if (!a) return 0;
        const int vr = (mm->vert_rate_sign == 0 ? 1 : -1) * (mm->vert_rate - 1) * 64;
        n = snprintf(buf, cap, "MSG,4,,,%s,,,,,,,,%d,%d,,,%i,,0,0,0,0", icao, a->speed, a->track, vr);
    } else
        return 0;
    if (n < 0 || (size_t)n + 2 > cap) return 0;
    buf[n++] = '\n';
    buf[n] = 0;
    return n;
}

}  // extern "C"
