// Sanitizer harness of the .ptau parser (keaki_amd/host/ptau.cpp, the mirror of the reference's src/kzg/ptau.rs:230-358): parses every file
// named on the command line and prints one line per file -- "ok <g1 points> <g2 points> <power>" or "err <kind>". Built with
// -fsanitize=address,undefined by `make ptau_fuzz_asan`; tests/test_ptau_asan.py feeds it truncated and corrupted copies of the reference's
// fixture: whatever the bytes, the parser must answer with a value or an error, never read outside the file. CPU only, no GPU library.
#include <cstdio>

#include "ptau.hpp"

int main(int argc, char** argv) {
  for (int i = 1; i < argc; i++) {
    auto r = keaki::ptau::get_powers_from_file(argv[i]);
    if (r.ok) printf("ok %zu %zu %u\n", r.value.g1.size(), r.value.g2.size(), (unsigned)r.value.header.power);
    else printf("err %d\n", (int)r.error.kind);
  }
  return 0;
}
