// Test scaffolding for tools/sanitize_host.sh tsan: the multi-threaded resolve against the sequential one on a
// record file (64-byte modes_record each), built together with the host sources under -fsanitize=thread.
#include <cstdio>
#include <cstring>
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
    std::vector<char> a(n * 64 + 64), b(n * 64 + 64);
    modes_host_config c{1, 0, 1, 0};
    for (int pieces : {2, 5, 16}) {
        modes_host *h0 = modes_host_create(&c), *h1 = modes_host_create(&c);
        uint64_t na = 0, nb = 0;
        const uint64_t m0 = modes_host_resolve_raw(h0, r.data(), n, nullptr, 0, a.data(), a.size(), &na);
        const uint64_t m1 = modes_host_resolve_raw_mt(h1, r.data(), n, b.data(), b.size(), &nb, -pieces);
        if (m0 != m1 || na != nb || memcmp(a.data(), b.data(), na) != 0) { fprintf(stderr, "mismatch with %d pieces\n", pieces); return 1; }
        modes_host_destroy(h0);
        modes_host_destroy(h1);
    }
    printf("mt resolve == sequential on %zu records\n", n);
    return 0;
}
