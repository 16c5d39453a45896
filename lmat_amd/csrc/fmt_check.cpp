// fmt_check -- test helper: reads float bit patterns (hex, one per line), prints put_float's text for each.
#include <cstdint>
#include <cstring>
#include <iostream>
#include "outfmt.hpp"
int main() {
    std::string line, out;
    while (std::getline(std::cin, line)) {
        const uint32_t u = (uint32_t)strtoul(line.c_str(), nullptr, 16);
        float f;
        memcpy(&f, &u, 4);
        out.clear();
        lmat::put_float(out, f);
        std::cout << out << "\n";
    }
    return 0;
}
