// tools/dump_stamps.cpp -- host-only dump of the product's stamp atlases / arena templates (no GPU needed) so that
// the CPU test-suite can compare them with the oracle's rendering.  Build: g++ -O1 -std=c++17 -o dump_stamps dump_stamps.cpp
#include <cstdio>

#include "../endless-memory-gym_amd/csrc/mg_stamps.hpp"

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    double agent_scale = atof(argv[1]);
    int N = atoi(argv[2]);
    FILE* f = fopen(argv[3], "wb");
    int radius;
    auto sprites = mg::build_agent_sprites(agent_scale, &radius);
    auto glyphs = mg::build_glyphs(0.25);
    auto templ = mg::build_mortar_templates(N, 0.25, 84);
    int hdr[4] = {sprites[0].w, radius, (int)glyphs.size(), N};
    fwrite(hdr, sizeof(int), 4, f);
    for (auto& s : sprites) fwrite(s.px.data(), 1, s.px.size(), f);
    for (auto& g : glyphs) {
        int d[2] = {g.w, g.h};
        fwrite(d, sizeof(int), 2, f);
        fwrite(g.px.data(), 1, g.px.size(), f);
    }
    fwrite(templ.data(), 1, templ.size(), f);
    if (argc >= 6) {  // coin_scale exit_scale: the spotlight family's coin and exit (closed, open) stamps
        mg::Stamp extra[3] = {mg::build_coin(atof(argv[4])), mg::build_exit(atof(argv[5]), false), mg::build_exit(atof(argv[5]), true)};
        for (auto& g : extra) {
            int d[2] = {g.w, g.h};
            fwrite(d, sizeof(int), 2, f);
            fwrite(g.px.data(), 1, g.px.size(), f);
        }
    }
    fclose(f);
    return 0;
}
