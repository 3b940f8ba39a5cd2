/* evc_io.h - C ABI of libevc_io.so: host-side input for the hot path (no GPU, no torch, no TensorFlow).
 *
 * Replaces, on the reference's side (cs/ = code_student_uniform/):
 *   tf.TFRecordReader                                     cs/readers.py:186-187
 *   tf.parse_single_sequence_example(context id/labels,
 *       sequence features as bytes)                       cs/readers.py:193-199
 *   tf.decode_raw(uint8) + reshape + resize_axis          cs/readers.py:146-174, :8-43
 *   tf.sparse_to_dense(labels, num_classes)               cs/readers.py:200-204
 * The features are handed over as uint8; Dequantize (cs/utils.py:22-25) and the zero padding of
 * frames >= num_frames are applied on the GPU by evc_l2norm_chunk_fwd (include/evc.h).
 *
 * All functions are thread-safe (per-thread error text) and return EVC_IO_OK or a negative code. */
#ifndef EVC_IO_H_
#define EVC_IO_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EVC_IO_OK 0
#define EVC_IO_ERR_ARG (-1)
#define EVC_IO_ERR_FILE (-2)
#define EVC_IO_ERR_FORMAT (-3)

const char* evc_io_last_error(void);

/* CRC32C (Castagnoli) and TFRecord's masked form ((crc >> 15 | crc << 17) + 0xa282ead8). */
uint32_t evc_crc32c(const uint8_t* data, int64_t n);
uint32_t evc_masked_crc32c(const uint8_t* data, int64_t n);

/* Walks one TFRecord file.  Returns the number of records (>= 0) or a negative error.  When offsets /
 * lengths are non-NULL the first max_records payload offsets and byte lengths are stored.  verify_crc != 0
 * checks both checksums of every record (tf.TFRecordReader does). */
int64_t evc_tfrecord_scan(const char* path, int64_t* offsets, int64_t* lengths, int64_t max_records, int verify_crc);

/* Parses ONE serialized tf.train.SequenceExample of the YouTube-8M frame-level data set.
 *   feature_names / feature_sizes [num_features]   e.g. {"rgb","audio"} / {1024,128}   (cs/readers.py:127-144)
 *   frames_out   [max_frames][sum(feature_sizes)] uint8: features concatenated per frame (cs/readers.py:232),
 *                truncated at max_frames, rows >= num_frames zero
 *   num_frames_out  min(frames in record, max_frames)                                   (cs/readers.py:168)
 *   labels_out   up to max_labels class indices ("labels" context feature), count in num_labels_out
 *   id_out       NUL-terminated video id (context "id"), at most id_cap - 1 bytes; may be NULL
 * Errors: a frame whose byte length differs from feature_sizes[i]; features with different frame counts
 * (tf.assert_equal, cs/readers.py:225); none of the features present. */
int evc_parse_yt8m_frame_example(const uint8_t* buf, int64_t len, const char* const* feature_names,
                                 const int32_t* feature_sizes, int num_features, int max_frames,
                                 uint8_t* frames_out, int32_t* num_frames_out, int64_t* labels_out,
                                 int max_labels, int32_t* num_labels_out, char* id_out, int id_cap);

/* Reads and parses `count` records of one file (payload offsets / lengths from evc_tfrecord_scan) into
 * batch-major buffers: frames_out [count][max_frames][row] uint8, num_frames_out [count] int32,
 * labels_multi_hot [count][num_classes] uint8 (0/1), ids_out [count][id_cap] chars (may be NULL). */
int evc_read_yt8m_frame_records(const char* path, const int64_t* offsets, const int64_t* lengths, int count,
                                const char* const* feature_names, const int32_t* feature_sizes, int num_features,
                                int max_frames, int num_classes, uint8_t* frames_out, int32_t* num_frames_out,
                                uint8_t* labels_multi_hot, char* ids_out, int id_cap);

/* Video-level (pre-aggregated) records: tf.train.Example with float features (cs/readers.py:53-113).
 *   features_out [sum(feature_sizes)] float32, features concatenated in feature_names order (:109-110).
 * Errors: a requested feature missing or of the wrong length (FixedLenFeature, :103-105). */
int evc_parse_yt8m_video_example(const uint8_t* buf, int64_t len, const char* const* feature_names,
                                 const int32_t* feature_sizes, int num_features, float* features_out,
                                 int64_t* labels_out, int max_labels, int32_t* num_labels_out, char* id_out,
                                 int id_cap);
int evc_read_yt8m_video_records(const char* path, const int64_t* offsets, const int64_t* lengths, int count,
                                const char* const* feature_names, const int32_t* feature_sizes, int num_features,
                                int num_classes, float* features_out, uint8_t* labels_multi_hot, char* ids_out,
                                int id_cap);

#ifdef __cplusplus
}
#endif
#endif /* EVC_IO_H_ */
