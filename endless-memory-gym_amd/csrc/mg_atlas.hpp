// mg_atlas.hpp -- host-side container that uploads stamps / templates / tables for the raster kernel.
#pragma once
#include <vector>

#include "mg_family.hpp"
#include "mg_raster.hpp"
#include "mg_stamps.hpp"

namespace mg {

class Atlas {
   public:
    Atlas() {
        memset(&tables_, 0, sizeof(tables_));
        static const uint8_t RGB[][3] = {
            {0, 0, 0},       {250, 204, 153}, {250, 250, 250}, {50, 50, 50},  {255, 255, 255}, {255, 0, 0},   {0, 255, 0},
            {0, 0, 255},     {255, 255, 0},   {255, 165, 0},   {50, 50, 50},  {120, 120, 120}, {116, 1, 113}, {255, 94, 14},
            {210, 210, 210}, {0, 0, 0},       {48, 141, 70},   {55, 55, 55},  {125, 177, 250}};
        for (int i = 0; i < (int)(sizeof(RGB) / 3); ++i) {
            tables_.palette[i] = (uint32_t)RGB[i][0] | ((uint32_t)RGB[i][1] << 8) | ((uint32_t)RGB[i][2] << 16);
            tables_.border_of[i] = (uint32_t)i;
        }
        tables_.border_of[C_WHITE] = C_GREY210;
        tables_.border_of[C_ICY] = C_GREY210;
    }

    // stamp pixels are palette ids (0 = transparent); stored column-major [x][y] as RGBA words with the column
    // stride padded to a power of two (pixel index p -> x = p >> sh, y = p & (stride - 1): no division in the kernel)
    // max_px: what the composer that draws this stamp can hold in registers (256 pixels per StampRegs slot); a scale
    // option that needs more is refused instead of being drawn truncated
    int add_stamp(const Stamp& s, int max_px = 1 << 30) {
        int id = n_stamps_++;
        if (id >= MAX_STAMPS) throw std::runtime_error("too many stamps");
        int sh = 0;
        while ((1 << sh) < s.h) ++sh;
        if ((s.w << sh) > max_px)
            throw OptionError{-3, "a *_scale reset parameter makes a sprite larger than the " + std::to_string(max_px) +
                                      " (padded) pixels this build's raster holds per layer"};
        tables_.stamps[id].off = (uint32_t)data_.size();
        tables_.stamps[id].w = (uint16_t)s.w;
        tables_.stamps[id].h = (uint16_t)s.h;
        tables_.stamps[id].sh = (uint16_t)sh;
        tables_.stamps[id].pad = 0;
        for (int x = 0; x < s.w; ++x)
            for (int y = 0; y < (1 << sh); ++y) data_.push_back(y < s.h ? rgba(s.get(x, y)) : 0u);
        return id;
    }
    // palette id -> r | g<<8 | b<<16 | 0xFF<<24 (opaque); id 0 is the colour key -> 0 (transparent)
    uint32_t rgba(uint8_t id) const { return id ? (tables_.palette[id] | 0xFF000000u) : 0u; }
    int n_stamps() const { return n_stamps_; }
    void set_templates(const std::vector<uint8_t>& t) { templates_ = t; }

    void upload() {
        if (data_.empty()) data_.push_back(0);
        if (templates_.empty()) templates_.resize(16, 0);
        stamp_dev_.upload(data_);
        templ_dev_.upload(templates_);
        // disc column spans for r = 0..DISC_RMAX
        std::vector<int8_t> span((size_t)(DISC_RMAX + 1) * 2 * DISC_RMAX * 2, 0);
        for (int r = 0; r <= DISC_RMAX; ++r) {
            Stamp s(2 * r + 2, 2 * r + 2);
            if (r >= 1) disc(s, r, r, r, 1);
            for (int i = 0; i < 2 * DISC_RMAX; ++i) {
                int lo = 1, hi = 0;
                if (i < 2 * r) {
                    int ymin = 1 << 20, ymax = -1;
                    for (int y = 0; y < s.h; ++y)
                        if (s.get(i, y)) {
                            ymin = std::min(ymin, y);
                            ymax = std::max(ymax, y);
                        }
                    if (ymax >= 0) {
                        lo = ymin - r;
                        hi = ymax - r;
                    }
                }
                span[((size_t)r * 2 * DISC_RMAX + i) * 2] = (int8_t)lo;
                span[((size_t)r * 2 * DISC_RMAX + i) * 2 + 1] = (int8_t)hi;
            }
        }
        span_dev_.upload(span);
        std::vector<AtlasTables> t(1, tables_);
        tables_dev_.upload(t);
        dev_.templates = templ_dev_.p;
        dev_.stamp_data = stamp_dev_.p;
        dev_.disc_span = span_dev_.p;
        dev_.tables = tables_dev_.p;
    }
    const RasterAtlas& dev() const { return dev_; }

   private:
    AtlasTables tables_;
    int n_stamps_ = 0;
    std::vector<uint32_t> data_;
    std::vector<uint8_t> templates_;
    DevArray<uint32_t> stamp_dev_;
    DevArray<uint8_t> templ_dev_;
    DevArray<int8_t> span_dev_;
    DevArray<AtlasTables> tables_dev_;
    RasterAtlas dev_;
};

}  // namespace mg
