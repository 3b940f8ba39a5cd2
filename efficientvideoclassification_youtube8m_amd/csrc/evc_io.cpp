// Host-side input library (libevc_io.so, plain C++17, no HIP): TFRecord framing and a
// dependency-free parser of YouTube-8M frame-level tf.train.SequenceExample records.
//
// Replaces tf.TFRecordReader + tf.parse_single_sequence_example + decode_raw/resize_axis of
// cs/readers.py:146-246 on the host; the features stay uint8 (4x less PCIe traffic than the
// reference's float32 feed) and are dequantised on the GPU by evc_l2norm_chunk_fwd.
//
// TFRecord record: uint64 length | uint32 masked_crc32c(length) | data | uint32 masked_crc32c(data).
// SequenceExample { Features context = 1; FeatureLists feature_lists = 2; }
//   Features     { map<string, Feature> feature = 1; }        (map entry: key = 1, value = 2)
//   FeatureLists { map<string, FeatureList> feature_list = 1; }
//   FeatureList  { repeated Feature feature = 1; }
//   Feature      { oneof { BytesList bytes_list = 1; FloatList float_list = 2; Int64List int64_list = 3; } }
//   BytesList    { repeated bytes value = 1; }   Int64List { repeated int64 value = 1 [packed]; }
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/evc_io.h"

static thread_local char g_io_err[512] = "";
#define IO_FAIL(code, ...)                               \
  do {                                                   \
    snprintf(g_io_err, sizeof(g_io_err), __VA_ARGS__);   \
    return (code);                                       \
  } while (0)

extern "C" const char* evc_io_last_error(void) { return g_io_err; }

// ---- CRC32C (Castagnoli), table driven ---------------------------------------------------------
static uint32_t g_crc_table[8][256];
static bool g_crc_init = false;
static void crc_init() {
  if (g_crc_init) return;
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);
    g_crc_table[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i)
    for (int t = 1; t < 8; ++t) g_crc_table[t][i] = (g_crc_table[t - 1][i] >> 8) ^ g_crc_table[0][g_crc_table[t - 1][i] & 0xff];
  g_crc_init = true;
}
extern "C" uint32_t evc_crc32c(const uint8_t* p, int64_t n) {
  crc_init();
  uint32_t c = 0xFFFFFFFFu;
  while (n >= 8) {
    uint64_t v;
    memcpy(&v, p, 8);
    v ^= c;
    c = g_crc_table[7][v & 0xff] ^ g_crc_table[6][(v >> 8) & 0xff] ^ g_crc_table[5][(v >> 16) & 0xff] ^
        g_crc_table[4][(v >> 24) & 0xff] ^ g_crc_table[3][(v >> 32) & 0xff] ^ g_crc_table[2][(v >> 40) & 0xff] ^
        g_crc_table[1][(v >> 48) & 0xff] ^ g_crc_table[0][(v >> 56) & 0xff];
    p += 8;
    n -= 8;
  }
  while (n-- > 0) c = g_crc_table[0][(c ^ *p++) & 0xff] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}
extern "C" uint32_t evc_masked_crc32c(const uint8_t* p, int64_t n) {
  const uint32_t c = evc_crc32c(p, n);
  return ((c >> 15) | (c << 17)) + 0xa282ead8u;
}

// ---- files ---------------------------------------------------------------------------------------
// Closed on every exit path, including an exception on its way to the catch clauses of the entry points below
// (no C++ exception crosses the extern "C" boundary: a corrupt length field becomes an error code, not an abort).
struct File {
  FILE* f;
  explicit File(const char* path) : f(fopen(path, "rb")) {}
  ~File() { if (f) fclose(f); }
  File(const File&) = delete;
  File& operator=(const File&) = delete;
  int64_t size() {
    const long cur = ftell(f);
    if (fseek(f, 0, SEEK_END) != 0) return -1;
    const long n = ftell(f);
    fseek(f, cur, SEEK_SET);
    return (int64_t)n;
  }
};
#define IO_GUARD_END(fn_)                                                                          \
  catch (const std::bad_alloc&) { IO_FAIL(EVC_IO_ERR_FORMAT, fn_ ": out of memory (corrupt length field?)"); } \
  catch (const std::exception& e) { IO_FAIL(EVC_IO_ERR_FORMAT, fn_ ": %s", e.what()); }

// ---- TFRecord scan --------------------------------------------------------------------------------
extern "C" int64_t evc_tfrecord_scan(const char* path, int64_t* offsets, int64_t* lengths, int64_t max_records, int verify_crc) try {
  File file(path);
  FILE* f = file.f;
  if (!f) IO_FAIL(EVC_IO_ERR_FILE, "cannot open %s", path);
  const int64_t fsize = file.size();
  int64_t count = 0, pos = 0;
  std::vector<uint8_t> buf;
  for (;;) {
    uint8_t hdr[12];
    const size_t got = fread(hdr, 1, 12, f);
    if (got == 0) break;
    if (got != 12) { IO_FAIL(EVC_IO_ERR_FORMAT, "%s: truncated record header at %lld", path, (long long)pos); }
    uint64_t len;
    uint32_t lcrc;
    memcpy(&len, hdr, 8);
    memcpy(&lcrc, hdr + 8, 4);
    if (verify_crc && evc_masked_crc32c(hdr, 8) != lcrc) { IO_FAIL(EVC_IO_ERR_FORMAT, "%s: bad length crc at %lld", path, (long long)pos); }
    // a record cannot be longer than what is left of the file (a corrupt length would otherwise size a buffer)
    if (fsize >= 0 && (len > (uint64_t)fsize || pos + 12 + (int64_t)len + 4 > fsize))
      IO_FAIL(EVC_IO_ERR_FORMAT, "%s: truncated record at %lld (claims %llu bytes, the file has %lld)", path, (long long)pos, (unsigned long long)len, (long long)fsize);
    if (offsets && count < max_records) { offsets[count] = pos + 12; lengths[count] = (int64_t)len; }
    if (verify_crc) {
      buf.resize(len + 4);
      if (fread(buf.data(), 1, len + 4, f) != len + 4) { IO_FAIL(EVC_IO_ERR_FORMAT, "%s: truncated record at %lld", path, (long long)pos); }
      uint32_t dcrc;
      memcpy(&dcrc, buf.data() + len, 4);
      if (evc_masked_crc32c(buf.data(), (int64_t)len) != dcrc) { IO_FAIL(EVC_IO_ERR_FORMAT, "%s: bad data crc at %lld", path, (long long)pos); }
    } else if (fseek(f, (long)(len + 4), SEEK_CUR) != 0) {
      IO_FAIL(EVC_IO_ERR_FORMAT, "%s: seek failed at %lld", path, (long long)pos);
    }
    pos += 12 + (int64_t)len + 4;
    ++count;
  }
  return count;
}
IO_GUARD_END("evc_tfrecord_scan")

// ---- protobuf wire helpers --------------------------------------------------------------------------
struct Span { const uint8_t* p; const uint8_t* e; };
static bool rd_varint(Span& s, uint64_t& v) {
  v = 0;
  for (int shift = 0; shift < 64 && s.p < s.e; shift += 7) {
    const uint8_t b = *s.p++;
    v |= (uint64_t)(b & 0x7f) << shift;
    if (!(b & 0x80)) return true;
  }
  return false;
}
static bool rd_len(Span& s, Span& sub) {
  uint64_t n;
  if (!rd_varint(s, n) || (uint64_t)(s.e - s.p) < n) return false;
  sub.p = s.p;
  sub.e = s.p + n;
  s.p += n;
  return true;
}
static bool skip_field(Span& s, int wt) {
  uint64_t v;
  Span sub;
  switch (wt) {
    case 0: return rd_varint(s, v);
    case 1: if (s.e - s.p < 8) return false; s.p += 8; return true;
    case 2: return rd_len(s, sub);
    case 5: if (s.e - s.p < 4) return false; s.p += 4; return true;
    default: return false;
  }
}
// map entry { key = 1 (string), value = 2 (message) }
static bool rd_map_entry(Span entry, std::string& key, Span& value) {
  bool have_v = false;
  key.clear();
  while (entry.p < entry.e) {
    uint64_t tag;
    if (!rd_varint(entry, tag)) return false;
    const int fn = (int)(tag >> 3), wt = (int)(tag & 7);
    if (fn == 1 && wt == 2) { Span k; if (!rd_len(entry, k)) return false; key.assign((const char*)k.p, k.e - k.p); }
    else if (fn == 2 && wt == 2) { if (!rd_len(entry, value)) return false; have_v = true; }
    else if (!skip_field(entry, wt)) return false;
  }
  return have_v;
}
// Feature -> first bytes value (bytes_list = 1 { value = 1 })
static bool feature_first_bytes(Span feat, Span& out) {
  while (feat.p < feat.e) {
    uint64_t tag;
    if (!rd_varint(feat, tag)) return false;
    const int fn = (int)(tag >> 3), wt = (int)(tag & 7);
    if (fn == 1 && wt == 2) {
      Span bl;
      if (!rd_len(feat, bl)) return false;
      while (bl.p < bl.e) {
        uint64_t t2;
        if (!rd_varint(bl, t2)) return false;
        if ((t2 >> 3) == 1 && (t2 & 7) == 2) return rd_len(bl, out);
        if (!skip_field(bl, (int)(t2 & 7))) return false;
      }
      return false;
    }
    if (!skip_field(feat, wt)) return false;
  }
  return false;
}
// Feature -> int64 values (int64_list = 3 { value = 1, packed or not })
static bool feature_int64s(Span feat, std::vector<int64_t>& out) {
  while (feat.p < feat.e) {
    uint64_t tag;
    if (!rd_varint(feat, tag)) return false;
    const int fn = (int)(tag >> 3), wt = (int)(tag & 7);
    if (fn == 3 && wt == 2) {
      Span il;
      if (!rd_len(feat, il)) return false;
      while (il.p < il.e) {
        uint64_t t2, v;
        if (!rd_varint(il, t2)) return false;
        if ((t2 >> 3) == 1 && (t2 & 7) == 0) { if (!rd_varint(il, v)) return false; out.push_back((int64_t)v); }
        else if ((t2 >> 3) == 1 && (t2 & 7) == 2) {
          Span pk;
          if (!rd_len(il, pk)) return false;
          while (pk.p < pk.e) { if (!rd_varint(pk, v)) return false; out.push_back((int64_t)v); }
        } else if (!skip_field(il, (int)(t2 & 7))) return false;
      }
      return true;
    }
    if (!skip_field(feat, wt)) return false;
  }
  return true;   // no int64_list: empty
}

extern "C" int evc_parse_yt8m_frame_example(const uint8_t* buf, int64_t len, const char* const* feature_names,
                                            const int32_t* feature_sizes, int num_features, int max_frames,
                                            uint8_t* frames_out, int32_t* num_frames_out, int64_t* labels_out,
                                            int max_labels, int32_t* num_labels_out, char* id_out, int id_cap) try {
  if (!buf || len <= 0 || num_features <= 0 || max_frames <= 0) IO_FAIL(EVC_IO_ERR_ARG, "evc_parse_yt8m_frame_example: bad arguments");
  int row = 0;
  std::vector<int> col0(num_features);
  std::vector<char> seen(num_features, 0);
  for (int i = 0; i < num_features; ++i) { col0[i] = row; row += feature_sizes[i]; }
  memset(frames_out, 0, (size_t)max_frames * row);      // resize_axis pads with zeros (cs/readers.py:8-43)
  if (id_out && id_cap > 0) id_out[0] = 0;
  *num_labels_out = 0;
  int num_frames = -1;
  Span ex{buf, buf + len};
  std::string key;
  while (ex.p < ex.e) {
    uint64_t tag;
    if (!rd_varint(ex, tag)) IO_FAIL(EVC_IO_ERR_FORMAT, "SequenceExample: bad tag");
    const int fn = (int)(tag >> 3), wt = (int)(tag & 7);
    if (fn == 1 && wt == 2) {            // context Features
      Span ctx;
      if (!rd_len(ex, ctx)) IO_FAIL(EVC_IO_ERR_FORMAT, "SequenceExample: bad context");
      while (ctx.p < ctx.e) {
        uint64_t t2;
        if (!rd_varint(ctx, t2)) IO_FAIL(EVC_IO_ERR_FORMAT, "context: bad tag");
        if ((t2 >> 3) == 1 && (t2 & 7) == 2) {
          Span entry, val;
          if (!rd_len(ctx, entry) || !rd_map_entry(entry, key, val)) IO_FAIL(EVC_IO_ERR_FORMAT, "context: bad map entry");
          if (key == "id" || key == "video_id") {          // cs/readers.py:195 reads "id" (2018 naming)
            Span b;
            if (feature_first_bytes(val, b) && id_out && id_cap > 0) {
              const int n = (int)std::min<int64_t>(b.e - b.p, id_cap - 1);
              memcpy(id_out, b.p, n);
              id_out[n] = 0;
            }
          } else if (key == "labels") {
            std::vector<int64_t> v;
            if (!feature_int64s(val, v)) IO_FAIL(EVC_IO_ERR_FORMAT, "context: bad labels");
            const int n = (int)std::min<size_t>(v.size(), (size_t)max_labels);
            for (int i = 0; i < n; ++i) labels_out[i] = v[i];
            *num_labels_out = n;
          }
        } else if (!skip_field(ctx, (int)(t2 & 7))) IO_FAIL(EVC_IO_ERR_FORMAT, "context: bad field");
      }
    } else if (fn == 2 && wt == 2) {     // FeatureLists
      Span fls;
      if (!rd_len(ex, fls)) IO_FAIL(EVC_IO_ERR_FORMAT, "SequenceExample: bad feature_lists");
      while (fls.p < fls.e) {
        uint64_t t2;
        if (!rd_varint(fls, t2)) IO_FAIL(EVC_IO_ERR_FORMAT, "feature_lists: bad tag");
        if ((t2 >> 3) == 1 && (t2 & 7) == 2) {
          Span entry, fl;
          if (!rd_len(fls, entry) || !rd_map_entry(entry, key, fl)) IO_FAIL(EVC_IO_ERR_FORMAT, "feature_lists: bad map entry");
          int fi = -1;
          for (int i = 0; i < num_features; ++i) if (key == feature_names[i]) fi = i;
          if (fi < 0) continue;
          seen[fi] = 1;
          int t = 0;
          while (fl.p < fl.e) {          // FeatureList: repeated Feature feature = 1
            uint64_t t3;
            if (!rd_varint(fl, t3)) IO_FAIL(EVC_IO_ERR_FORMAT, "feature_list: bad tag");
            if ((t3 >> 3) == 1 && (t3 & 7) == 2) {
              Span feat, b;
              if (!rd_len(fl, feat)) IO_FAIL(EVC_IO_ERR_FORMAT, "feature_list: bad feature");
              if (!feature_first_bytes(feat, b)) IO_FAIL(EVC_IO_ERR_FORMAT, "feature '%s' frame %d: no bytes value", key.c_str(), t);
              if (b.e - b.p != feature_sizes[fi]) IO_FAIL(EVC_IO_ERR_FORMAT, "feature '%s' frame %d: %ld bytes, expected %d", key.c_str(), t, (long)(b.e - b.p), feature_sizes[fi]);
              if (t < max_frames) memcpy(frames_out + (size_t)t * row + col0[fi], b.p, feature_sizes[fi]);   // truncate at max_frames
              ++t;
            } else if (!skip_field(fl, (int)(t3 & 7))) IO_FAIL(EVC_IO_ERR_FORMAT, "feature_list: bad field");
          }
          const int nf = t < max_frames ? t : max_frames;       // tf.minimum(shape[0], max_frames) cs/readers.py:168
          if (num_frames == -1) num_frames = nf;
          else if (num_frames != nf) IO_FAIL(EVC_IO_ERR_FORMAT, "features disagree on the number of frames (%d vs %d)", num_frames, nf);   // tf.assert_equal :225
        } else if (!skip_field(fls, (int)(t2 & 7))) IO_FAIL(EVC_IO_ERR_FORMAT, "feature_lists: bad field");
      }
    } else if (!skip_field(ex, wt)) IO_FAIL(EVC_IO_ERR_FORMAT, "SequenceExample: bad field");
  }
  // tf.parse_single_sequence_example fails on a record that lacks any of the requested feature lists
  // (FixedLenSequenceFeature without allow_missing, cs/readers.py:196-199): no silently zero columns
  for (int i = 0; i < num_features; ++i)
    if (!seen[i]) IO_FAIL(EVC_IO_ERR_FORMAT, "feature list '%s' is missing from the record", feature_names[i]);
  if (num_frames < 0) IO_FAIL(EVC_IO_ERR_FORMAT, "none of the requested features is present in the record");
  *num_frames_out = num_frames;
  return EVC_IO_OK;
}
IO_GUARD_END("evc_parse_yt8m_frame_example")

// Feature -> float values (float_list = 2 { value = 1, packed or not })
static bool feature_floats(Span feat, std::vector<float>& out) {
  while (feat.p < feat.e) {
    uint64_t tag;
    if (!rd_varint(feat, tag)) return false;
    const int fn = (int)(tag >> 3), wt = (int)(tag & 7);
    if (fn == 2 && wt == 2) {
      Span fl;
      if (!rd_len(feat, fl)) return false;
      while (fl.p < fl.e) {
        uint64_t t2;
        if (!rd_varint(fl, t2)) return false;
        if ((t2 >> 3) == 1 && (t2 & 7) == 5) {
          if (fl.e - fl.p < 4) return false;
          float v; memcpy(&v, fl.p, 4); fl.p += 4; out.push_back(v);
        } else if ((t2 >> 3) == 1 && (t2 & 7) == 2) {
          Span pk;
          if (!rd_len(fl, pk) || ((pk.e - pk.p) & 3)) return false;
          const size_t n = (pk.e - pk.p) / 4, o = out.size();
          out.resize(o + n);
          memcpy(out.data() + o, pk.p, n * 4);
        } else if (!skip_field(fl, (int)(t2 & 7))) return false;
      }
      return true;
    }
    if (!skip_field(feat, wt)) return false;
  }
  return true;
}

// tf.train.Example { Features features = 1; } of the video-level (pre-aggregated) data set.
extern "C" int evc_parse_yt8m_video_example(const uint8_t* buf, int64_t len, const char* const* feature_names,
                                            const int32_t* feature_sizes, int num_features, float* features_out,
                                            int64_t* labels_out, int max_labels, int32_t* num_labels_out, char* id_out,
                                            int id_cap) try {
  if (!buf || len <= 0 || num_features <= 0) IO_FAIL(EVC_IO_ERR_ARG, "evc_parse_yt8m_video_example: bad arguments");
  std::vector<int> col0(num_features);
  std::vector<char> seen(num_features, 0);
  int row = 0;
  for (int i = 0; i < num_features; ++i) { col0[i] = row; row += feature_sizes[i]; }
  if (id_out && id_cap > 0) id_out[0] = 0;
  *num_labels_out = 0;
  Span ex{buf, buf + len};
  std::string key;
  std::vector<float> vals;
  while (ex.p < ex.e) {
    uint64_t tag;
    if (!rd_varint(ex, tag)) IO_FAIL(EVC_IO_ERR_FORMAT, "Example: bad tag");
    if ((tag >> 3) == 1 && (tag & 7) == 2) {
      Span fs;
      if (!rd_len(ex, fs)) IO_FAIL(EVC_IO_ERR_FORMAT, "Example: bad features");
      while (fs.p < fs.e) {
        uint64_t t2;
        if (!rd_varint(fs, t2)) IO_FAIL(EVC_IO_ERR_FORMAT, "features: bad tag");
        if ((t2 >> 3) == 1 && (t2 & 7) == 2) {
          Span entry, val;
          if (!rd_len(fs, entry) || !rd_map_entry(entry, key, val)) IO_FAIL(EVC_IO_ERR_FORMAT, "features: bad map entry");
          if (key == "id" || key == "video_id") {
            Span b;
            if (feature_first_bytes(val, b) && id_out && id_cap > 0) {
              const int n = (int)std::min<int64_t>(b.e - b.p, id_cap - 1);
              memcpy(id_out, b.p, n);
              id_out[n] = 0;
            }
          } else if (key == "labels") {
            std::vector<int64_t> v;
            if (!feature_int64s(val, v)) IO_FAIL(EVC_IO_ERR_FORMAT, "features: bad labels");
            const int n = (int)std::min<size_t>(v.size(), (size_t)max_labels);
            for (int i = 0; i < n; ++i) labels_out[i] = v[i];
            *num_labels_out = n;
          } else {
            for (int i = 0; i < num_features; ++i) {
              if (key != feature_names[i]) continue;
              vals.clear();
              if (!feature_floats(val, vals) || (int)vals.size() != feature_sizes[i])      // FixedLenFeature([size], float32) cs/readers.py:103-105
                IO_FAIL(EVC_IO_ERR_FORMAT, "feature '%s': %zu floats, expected %d", key.c_str(), vals.size(), feature_sizes[i]);
              memcpy(features_out + col0[i], vals.data(), vals.size() * 4);
              seen[i] = 1;
            }
          }
        } else if (!skip_field(fs, (int)(t2 & 7))) IO_FAIL(EVC_IO_ERR_FORMAT, "features: bad field");
      }
    } else if (!skip_field(ex, (int)(tag & 7))) IO_FAIL(EVC_IO_ERR_FORMAT, "Example: bad field");
  }
  for (int i = 0; i < num_features; ++i)
    if (!seen[i]) IO_FAIL(EVC_IO_ERR_FORMAT, "feature '%s' is missing from the record", feature_names[i]);
  return EVC_IO_OK;
}
IO_GUARD_END("evc_parse_yt8m_video_example")

extern "C" int evc_read_yt8m_video_records(const char* path, const int64_t* offsets, const int64_t* lengths, int count,
                                           const char* const* feature_names, const int32_t* feature_sizes, int num_features,
                                           int num_classes, float* features_out, uint8_t* labels_multi_hot, char* ids_out,
                                           int id_cap) try {
  File file(path);
  FILE* f = file.f;
  if (!f) IO_FAIL(EVC_IO_ERR_FILE, "cannot open %s", path);
  const int64_t fsize = file.size();
  int row = 0;
  for (int i = 0; i < num_features; ++i) row += feature_sizes[i];
  std::vector<uint8_t> buf;
  std::vector<int64_t> labels(4096);
  for (int r = 0; r < count; ++r) {
    if (lengths[r] < 0 || offsets[r] < 0 || (fsize >= 0 && offsets[r] + lengths[r] > fsize))
      IO_FAIL(EVC_IO_ERR_FORMAT, "%s: record %d (offset %lld, %lld bytes) lies outside the file", path, r, (long long)offsets[r], (long long)lengths[r]);
    buf.resize(lengths[r]);
    if (fseek(f, (long)offsets[r], SEEK_SET) != 0 || fread(buf.data(), 1, lengths[r], f) != (size_t)lengths[r])
      IO_FAIL(EVC_IO_ERR_FORMAT, "%s: cannot read record %d", path, r);
    int32_t nl = 0;
    const int rc = evc_parse_yt8m_video_example(buf.data(), lengths[r], feature_names, feature_sizes, num_features,
                                                features_out + (size_t)r * row, labels.data(), (int)labels.size(), &nl,
                                                ids_out ? ids_out + (size_t)r * id_cap : nullptr, id_cap);
    if (rc != EVC_IO_OK) return rc;
    uint8_t* mh = labels_multi_hot + (size_t)r * num_classes;   // tf.sparse_to_indicator cs/readers.py:108
    memset(mh, 0, num_classes);
    for (int i = 0; i < nl; ++i)
      if (labels[i] >= 0 && labels[i] < num_classes) mh[labels[i]] = 1;
  }
  return EVC_IO_OK;
}
IO_GUARD_END("evc_read_yt8m_records")

// Reads `count` records (given by offset/length) of one file and parses them into batch buffers.
extern "C" int evc_read_yt8m_frame_records(const char* path, const int64_t* offsets, const int64_t* lengths, int count,
                                           const char* const* feature_names, const int32_t* feature_sizes, int num_features,
                                           int max_frames, int num_classes, uint8_t* frames_out, int32_t* num_frames_out,
                                           uint8_t* labels_multi_hot, char* ids_out, int id_cap) try {
  File file(path);
  FILE* f = file.f;
  if (!f) IO_FAIL(EVC_IO_ERR_FILE, "cannot open %s", path);
  const int64_t fsize = file.size();
  int row = 0;
  for (int i = 0; i < num_features; ++i) row += feature_sizes[i];
  std::vector<uint8_t> buf;
  std::vector<int64_t> labels(4096);
  for (int r = 0; r < count; ++r) {
    if (lengths[r] < 0 || offsets[r] < 0 || (fsize >= 0 && offsets[r] + lengths[r] > fsize))
      IO_FAIL(EVC_IO_ERR_FORMAT, "%s: record %d (offset %lld, %lld bytes) lies outside the file", path, r, (long long)offsets[r], (long long)lengths[r]);
    buf.resize(lengths[r]);
    if (fseek(f, (long)offsets[r], SEEK_SET) != 0 || fread(buf.data(), 1, lengths[r], f) != (size_t)lengths[r])
      IO_FAIL(EVC_IO_ERR_FORMAT, "%s: cannot read record %d", path, r);
    int32_t nl = 0;
    const int rc = evc_parse_yt8m_frame_example(buf.data(), lengths[r], feature_names, feature_sizes, num_features, max_frames,
                                                frames_out + (size_t)r * max_frames * row, num_frames_out + r, labels.data(),
                                                (int)labels.size(), &nl, ids_out ? ids_out + (size_t)r * id_cap : nullptr, id_cap);
    if (rc != EVC_IO_OK) return rc;
    uint8_t* mh = labels_multi_hot + (size_t)r * num_classes;   // tf.sparse_to_dense(labels, (num_classes,), 1) cs/readers.py:200-204
    memset(mh, 0, num_classes);
    for (int i = 0; i < nl; ++i)
      if (labels[i] >= 0 && labels[i] < num_classes) mh[labels[i]] = 1;
  }
  return EVC_IO_OK;
}
IO_GUARD_END("evc_read_yt8m_records")
