// Native (host) reader of the reference's dataset files: GZIP TFRecord framing + tf.train.SequenceExample decoding +
// the loader's per-record decode.  No device code; caller-owned memory throughout; zlib for the GZIP stream.
//
//   files written by convert_data.py:247-279 (TFRecordWriter, compression GZIP, one SequenceExample per record)
//   read by dataloader/outdoor_data_mfcc.py:62 (TFRecordDataset(..., compression_type='GZIP')) and parsed by
//   `_parse_sequence` :263-343 (context int64 scalars; feature lists of raw-bytes steps; tf.decode_raw; the acoustic
//   image flipped left-right and up-down :314-315).
//
// TFRecord framing (record_writer.cc): [u64 length][u32 masked crc32c(length)][data][u32 masked crc32c(data)],
// masked = rotr(crc, 15) + 0xa282ead8.  Protobuf wire format decoded by hand (varint / 64-bit / length-delimited /
// 32-bit); SequenceExample { Features context = 1; FeatureLists feature_lists = 2 }, both map<string, ...> fields
// (entry: key = 1, value = 2); Feature { BytesList = 1 | FloatList = 2 | Int64List = 3 }, each { repeated value = 1 }.
#include <zlib.h>

#include "common.hpp"

using namespace acimg;

extern "C" uint32_t acimg_crc32c(const void* data, size_t n, uint32_t crc);

namespace {

inline uint32_t unmask(uint32_t m) {
    const uint32_t rot = m - 0xa282ead8u;
    return (rot >> 17) | (rot << 15);
}
inline uint32_t rd32(const uint8_t* p) {
    uint32_t v;
    memcpy(&v, p, 4);
    return v;
}
inline uint64_t rd64(const uint8_t* p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;
}

struct Span {
    const uint8_t* p;
    size_t n;
};

// one protobuf field: returns false at the end of the span or on malformed input (ok = false then)
struct Field {
    uint32_t num, wt;
    uint64_t val;   // varint / fixed value
    Span sub;       // length-delimited payload
};
inline bool varint(const uint8_t*& p, const uint8_t* end, uint64_t& v) {
    v = 0;
    for (int shift = 0; shift < 70 && p < end; shift += 7) {
        const uint8_t b = *p++;
        v |= (uint64_t)(b & 0x7f) << shift;
        if (!(b & 0x80)) return true;
    }
    return false;
}
inline bool next_field(const uint8_t*& p, const uint8_t* end, Field& f, bool& ok) {
    if (p >= end) return false;
    uint64_t key;
    if (!varint(p, end, key)) { ok = false; return false; }
    f.num = (uint32_t)(key >> 3);
    f.wt = (uint32_t)(key & 7);
    f.val = 0;
    f.sub = Span{nullptr, 0};
    switch (f.wt) {
        case 0:
            if (!varint(p, end, f.val)) { ok = false; return false; }
            return true;
        case 1:
            if (end - p < 8) { ok = false; return false; }
            f.val = rd64(p);
            p += 8;
            return true;
        case 2: {
            uint64_t len;
            if (!varint(p, end, len) || len > (uint64_t)(end - p)) { ok = false; return false; }
            f.sub = Span{p, (size_t)len};
            p += len;
            return true;
        }
        case 5:
            if (end - p < 4) { ok = false; return false; }
            f.val = rd32(p);
            p += 4;
            return true;
        default:
            ok = false;
            return false;
    }
}

// map entry {key = 1 (string), value = 2 (message)}
inline bool map_entry(Span e, Span& key, Span& val) {
    const uint8_t* p = e.p;
    const uint8_t* end = e.p + e.n;
    Field f;
    bool ok = true;
    key = Span{nullptr, 0};
    val = Span{nullptr, 0};
    while (next_field(p, end, f, ok)) {
        if (f.num == 1 && f.wt == 2) key = f.sub;
        else if (f.num == 2 && f.wt == 2) val = f.sub;
    }
    return ok;
}

inline bool key_is(Span k, const char* s) { return k.n == strlen(s) && memcmp(k.p, s, k.n) == 0; }

// first int64 of a Feature (Int64List, packed or not); false when the feature holds none; `ok` goes false on a
// malformed sub-message (the caller refuses the record)
inline bool feature_int64(Span feat, int64_t& out, bool& ok) {
    const uint8_t* p = feat.p;
    const uint8_t* end = feat.p + feat.n;
    Field f;
    while (next_field(p, end, f, ok)) {
        if (f.num != 3 || f.wt != 2) continue;
        const uint8_t* q = f.sub.p;
        const uint8_t* qe = q + f.sub.n;
        Field g;
        while (next_field(q, qe, g, ok)) {
            if (g.num != 1) continue;
            if (g.wt == 0) { out = (int64_t)g.val; return true; }
            if (g.wt == 2) {
                const uint8_t* r = g.sub.p;
                uint64_t v;
                if (varint(r, r + g.sub.n, v)) { out = (int64_t)v; return true; }
            }
        }
    }
    return false;
}
// the single bytes value of a Feature (BytesList with one element: _bytes_feature of convert_data.py)
inline bool feature_bytes(Span feat, Span& out) {
    const uint8_t* p = feat.p;
    const uint8_t* end = feat.p + feat.n;
    Field f;
    bool ok = true;
    while (next_field(p, end, f, ok)) {
        if (f.num != 1 || f.wt != 2) continue;
        const uint8_t* q = f.sub.p;
        const uint8_t* qe = q + f.sub.n;
        Field g;
        while (next_field(q, qe, g, ok))
            if (g.num == 1 && g.wt == 2) { out = g.sub; return true; }
    }
    return false;
}

}  // namespace

extern "C" {

int acimg_gzip_inflate(const uint8_t* src, size_t src_len, uint8_t* out, size_t cap, size_t* produced) {
    if (!src || !produced) return fail(ACIMG_EINVAL, "gzip_inflate: null argument");
    *produced = 0;
    if (src_len < 2 || src[0] != 0x1f || src[1] != 0x8b) {       // not GZIP: the file is the record stream itself
        *produced = src_len;
        if (!out || cap < src_len) return fail(ACIMG_EWORKSPACE, "gzip_inflate: output %zu < %zu", cap, src_len);
        memcpy(out, src, src_len);
        return ACIMG_OK;
    }
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, 15 + 16) != Z_OK) return fail(ACIMG_ELAUNCH, "gzip_inflate: inflateInit2 failed");
    uint8_t scratch[1 << 15];
    size_t total = 0;
    bool fits = true;
    zs.next_in = const_cast<Bytef*>(src);
    size_t in_left = src_len;
    int rc = Z_OK;
    for (;;) {
        if (zs.avail_in == 0 && in_left) {
            const size_t take = in_left > (1u << 30) ? (1u << 30) : in_left;
            zs.avail_in = (uInt)take;
            in_left -= take;
        }
        uint8_t* dst = (fits && out && total < cap) ? out + total : scratch;
        const size_t room = (dst == scratch) ? sizeof(scratch) : (cap - total > (1u << 30) ? (1u << 30) : cap - total);
        zs.next_out = dst;
        zs.avail_out = (uInt)room;
        rc = inflate(&zs, Z_NO_FLUSH);
        const size_t got = room - zs.avail_out;
        if (dst == scratch && got) fits = false;       // counted, not stored: the caller learns the size it needs
        total += got;
        if (rc == Z_STREAM_END) {
            if (zs.avail_in || in_left) {              // concatenated members (a TFRecord writer emits one, gzip allows more)
                if (inflateReset(&zs) != Z_OK) break;
                continue;
            }
            break;
        }
        if (rc != Z_OK && rc != Z_BUF_ERROR) break;
        if (rc == Z_BUF_ERROR && zs.avail_in == 0 && in_left == 0) break;   // truncated stream
    }
    inflateEnd(&zs);
    *produced = total;
    if (rc != Z_STREAM_END) return fail(ACIMG_EINVAL, "gzip_inflate: corrupt or truncated GZIP stream (zlib %d)", rc);
    if (!fits || !out || total > cap) return fail(ACIMG_EWORKSPACE, "gzip_inflate: output %zu < %zu", cap, total);
    return ACIMG_OK;
}

long acimg_tfrecord_index(const uint8_t* buf, size_t len, uint64_t* offsets, uint64_t* lengths, long cap, int verify) {
    if (!buf && len) return fail(ACIMG_EINVAL, "tfrecord_index: null buffer");
    size_t pos = 0;
    long n = 0;
    while (pos < len) {
        if (len - pos < 12) return fail(ACIMG_EINVAL, "tfrecord_index: truncated TFRecord header at byte %zu", pos);
        const uint64_t dl = rd64(buf + pos);
        if (verify && unmask(rd32(buf + pos + 8)) != acimg_crc32c(buf + pos, 8, 0))
            return fail(ACIMG_EINVAL, "tfrecord_index: length checksum mismatch at byte %zu", pos);
        if (dl > len - pos - 12 || len - pos - 12 - dl < 4)
            return fail(ACIMG_EINVAL, "tfrecord_index: truncated TFRecord (record %ld wants %llu bytes)", n,
                        (unsigned long long)dl);
        const uint8_t* data = buf + pos + 12;
        if (verify && unmask(rd32(data + dl)) != acimg_crc32c(data, (size_t)dl, 0))
            return fail(ACIMG_EINVAL, "tfrecord_index: payload checksum mismatch in record %ld", n);
        if (n < cap && offsets && lengths) {
            offsets[n] = pos + 12;
            lengths[n] = dl;
        }
        ++n;
        pos += 12 + (size_t)dl + 4;
    }
    return n;
}

int acimg_sequence_example_decode(const uint8_t* rec, size_t len, AcimgSequenceDims* dims, float* audio_images,
                                  size_t audio_images_cap, int32_t* audio_samples, size_t audio_samples_cap,
                                  uint8_t* video, size_t video_cap) {
    if (!rec || !dims) return fail(ACIMG_EINVAL, "sequence_example_decode: null argument");
    memset(dims, 0, sizeof(*dims));
    dims->classes = dims->location = -1;
    Span lists[3] = {{nullptr, 0}, {nullptr, 0}, {nullptr, 0}};      // audio/image, audio/data, video/image
    const uint8_t* p = rec;
    const uint8_t* end = rec + len;
    Field f;
    bool ok = true;
    while (next_field(p, end, f, ok)) {
        if (f.wt != 2 || (f.num != 1 && f.num != 2)) continue;
        const uint8_t* q = f.sub.p;
        const uint8_t* qe = q + f.sub.n;
        Field e;
        while (next_field(q, qe, e, ok)) {
            if (e.num != 1 || e.wt != 2) continue;
            Span key, val;
            if (!map_entry(e.sub, key, val)) { ok = false; break; }
            if (f.num == 1) {                                        // context: int64 scalars
                int64_t v;
                if (!feature_int64(val, v, ok)) continue;
                if (key_is(key, "classes")) dims->classes = v;
                else if (key_is(key, "location")) dims->location = v;
                else if (key_is(key, "audio_image/height")) dims->audio_height = v;
                else if (key_is(key, "audio_image/width")) dims->audio_width = v;
                else if (key_is(key, "audio_image/depth")) dims->audio_depth = v;
                else if (key_is(key, "audio_data/mics")) dims->mics = v;
                else if (key_is(key, "audio_data/samples")) dims->samples = v;
                else if (key_is(key, "video/height")) dims->video_height = v;
                else if (key_is(key, "video/width")) dims->video_width = v;
                else if (key_is(key, "video/depth")) dims->video_depth = v;
            } else {                                                 // feature lists
                if (key_is(key, "audio/image")) lists[0] = val;
                else if (key_is(key, "audio/data")) lists[1] = val;
                else if (key_is(key, "video/image")) lists[2] = val;
            }
        }
    }
    if (!ok) return fail(ACIMG_EINVAL, "sequence_example_decode: malformed protobuf");
    // The context dimensions come from the file: a dimension that is present must be a sane positive size BEFORE any
    // product is formed (H = W = -1 would overflow into a plausible byte count; a direct C-ABI caller has no NumPy
    // allocation failure to save it).  1 <= d <= 65536 keeps every product below 2^48 * 4.
    {
        const int64_t* dimv[8] = {&dims->audio_height, &dims->audio_width, &dims->audio_depth, &dims->mics, &dims->samples,
                                  &dims->video_height, &dims->video_width, &dims->video_depth};
        for (int i = 0; i < 8; ++i)
            if (*dimv[i] != 0 && (*dimv[i] < 1 || *dimv[i] > (i == 4 ? (int64_t)1 << 24 : 65536)))   // samples: 2^24
                return fail(ACIMG_EINVAL, "sequence_example_decode: context dimension %d = %lld is out of range", i,
                            (long long)*dimv[i]);
    }
    // steps of every list; bytes per step must match the context dimensions (tf.reshape would fail otherwise)
    int64_t* steps_out[3] = {&dims->audio_image_steps, &dims->audio_data_steps, &dims->video_steps};
    const int64_t step_bytes[3] = {dims->audio_height * dims->audio_width * dims->audio_depth * 4, 0,
                                   dims->video_height * dims->video_width * dims->video_depth};
    for (int li = 0; li < 3; ++li) {
        if (!lists[li].p) continue;
        const uint8_t* q = lists[li].p;
        const uint8_t* qe = q + lists[li].n;
        Field s;
        int64_t step = 0;
        while (next_field(q, qe, s, ok)) {
            if (s.num != 1 || s.wt != 2) continue;
            Span raw;
            if (!feature_bytes(s.sub, raw)) return fail(ACIMG_EINVAL, "sequence_example_decode: step %lld holds no bytes", (long long)step);
            if (li == 0) {
                if (step_bytes[0] <= 0 || (int64_t)raw.n != step_bytes[0])
                    return fail(ACIMG_EINVAL, "sequence_example_decode: audio/image step of %zu bytes, context says %lld",
                                raw.n, (long long)step_bytes[0]);
                if (audio_images) {
                    const int64_t H = dims->audio_height, W = dims->audio_width, D = dims->audio_depth;
                    if (step >= (int64_t)1 << 24 || (size_t)((step + 1) * H * W * D) > audio_images_cap)
                        return fail(ACIMG_EWORKSPACE, "sequence_example_decode: audio_images buffer too small");
                    // tf.image.flip_left_right then flip_up_down (:314-315): out[h][w] = in[H-1-h][W-1-w]
                    float* dst = audio_images + step * H * W * D;
                    for (int64_t h = 0; h < H; ++h)
                        for (int64_t w = 0; w < W; ++w)
                            memcpy(dst + (h * W + w) * D, raw.p + ((H - 1 - h) * W + (W - 1 - w)) * D * 4, (size_t)D * 4);
                }
            } else if (li == 1) {
                if (raw.n % 4) return fail(ACIMG_EINVAL, "sequence_example_decode: audio/data step is not int32");
                if (dims->samples > 0 && (raw.n / 4) % (size_t)dims->samples)
                    return fail(ACIMG_EINVAL, "sequence_example_decode: audio/data step of %zu values, samples = %lld",
                                raw.n / 4, (long long)dims->samples);
                if (audio_samples) {
                    if ((size_t)dims->audio_data_values + raw.n / 4 > audio_samples_cap)
                        return fail(ACIMG_EWORKSPACE, "sequence_example_decode: audio_samples buffer too small");
                    memcpy(audio_samples + dims->audio_data_values, raw.p, raw.n);
                }
                dims->audio_data_values += (int64_t)(raw.n / 4);
            } else {
                if (step_bytes[2] <= 0 || (int64_t)raw.n != step_bytes[2])
                    return fail(ACIMG_EINVAL, "sequence_example_decode: video/image step of %zu bytes, context says %lld",
                                raw.n, (long long)step_bytes[2]);
                if (video) {
                    if (step >= (int64_t)1 << 24 || (size_t)((step + 1) * step_bytes[2]) > video_cap)
                        return fail(ACIMG_EWORKSPACE, "sequence_example_decode: video buffer too small");
                    memcpy(video + step * step_bytes[2], raw.p, raw.n);
                }
            }
            ++step;
        }
        if (!ok) return fail(ACIMG_EINVAL, "sequence_example_decode: malformed feature list");
        *steps_out[li] = step;
    }
    if (dims->classes < 0 || dims->location < 0)
        return fail(ACIMG_EINVAL, "sequence_example_decode: context lacks 'classes' / 'location' (FixedLenFeature)");
    return ACIMG_OK;
}

}  // extern "C"
