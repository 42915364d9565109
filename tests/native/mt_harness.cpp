// Test scaffolding for tools/sanitize_host.sh tsan: the multi-threaded resolve against the sequential one on a
// record file (64-byte modes_record each), built together with the host sources under -fsanitize=thread.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "../../include/modes_gfx950.h"
#include "../../include/modes_host.h"
int main(int argc, char **argv) {
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END);
    const size_t n = (size_t)ftell(f) / sizeof(modes_record);
    fseek(f, 0, SEEK_SET);
    std::vector<modes_record> r(n);
    if (fread(r.data(), sizeof(modes_record), n, f) != n) return 2;
    fclose(f);
    // two resolver threads at once (two hosts of one process: their parallel loops take turns, their pieces are their own), each
    // comparing the sequential listing with the multi-threaded one - gathered into one buffer, and left in its pieces
    // (modes_host_resolve_raw_pieces) - twice per host: the second call of a host that was caught with a wrong answer guesses
    std::atomic<int> bad{0};
    auto body = [&](int who) {
        std::vector<char> a(n * 64 + 64), b(n * 64 + 64);
        modes_host_config c{1, 0, 1, 0};
        for (int pieces : {2, 5, 16}) {
            modes_host *h0 = modes_host_create(&c), *h1 = modes_host_create(&c), *h2 = modes_host_create(&c);
            for (int call = 0; call < 2; call++) {
                uint64_t na = 0, nb = 0, nc = 0;
                const uint64_t m0 = modes_host_resolve_raw(h0, r.data(), n, nullptr, 0, a.data(), a.size(), &na);
                const uint64_t m1 = modes_host_resolve_raw_mt(h1, r.data(), n, b.data(), b.size(), &nb, -pieces);
                if (m0 != m1 || na != nb || memcmp(a.data(), b.data(), na) != 0) { fprintf(stderr, "thread %d: mismatch with %d pieces\n", who, pieces); bad++; }
                modes_text_piece tp[80];
                uint32_t np = 0;
                const modes_record *seg = r.data();
                const uint64_t nseg = n;
                const uint64_t m2 = modes_host_resolve_raw_pieces(h2, &seg, &nseg, 1, tp, 80, &np, &nc, -pieces);
                std::string joined;
                for (uint32_t i = 0; i < np; i++) joined.append(tp[i].base, (size_t)tp[i].len);
                if (m2 != m0 || nc != na || joined.size() != na || memcmp(joined.data(), a.data(), na) != 0) { fprintf(stderr, "thread %d: pieces differ (%d)\n", who, pieces); bad++; }
            }
            modes_host_destroy(h0);
            modes_host_destroy(h1);
            modes_host_destroy(h2);
        }
    };
    std::thread other(body, 1);
    body(0);
    other.join();
    if (bad.load()) return 1;
    printf("mt resolve == sequential on %zu records\n", n);
    return 0;
}
