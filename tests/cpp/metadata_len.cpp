// Prints what the host mirror would put in front of a ciphertext's polynomials (MetaDataJSON) and a few
// go-humanize renderings: tests/test_host_mirror.py holds them to the reference's size logs on CPU.
#include <cstdio>
#include <cstdlib>

#include "../../lumenos_amd/host/fhe.hpp"

int main(int argc, char **argv) {
    using namespace lumenos::fhe;
    MetaData md;
    md.Scale = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
    md.LogCols = argc > 2 ? atoi(argv[2]) : 11;
    const std::string s = MetaDataJSON(md, 144115188075593729ull);
    printf("%zu\n%s\n", s.size(), s.c_str());
    for (int i = 3; i < argc; i++) printf("%s\n", HumanizeBytes(strtoull(argv[i], nullptr, 10)).c_str());
    return 0;
}
