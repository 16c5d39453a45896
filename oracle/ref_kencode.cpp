// ref_kencode.cpp -- golden vectors from the reference's 2-bit k-mer encoder
// (include/kencode.hpp:79-85, forward strand only).  TEST INFRASTRUCTURE ONLY;
// built by oracle/Makefile against /root/reference into oracle/_ref/.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <kencode.hpp>

int main(int argc, char** argv) {
    // ref_kencode <k> <file of k-mer strings>
    if (argc < 3) return 2;
    int k = atoi(argv[1]);
    kencode_ns::kencode_c ken(k);
    std::ifstream in(argv[2]);
    std::string s;
    while (in >> s) printf("%s %llu\n", s.c_str(), (unsigned long long)ken.kencode(s.c_str()));
    return 0;
}
