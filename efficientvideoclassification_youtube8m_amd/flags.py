"""The reference's command-line flag surface.

Same names, defaults and syntax as the tf.flags definitions scattered over
cs/train.py:27-99, cs/frame_level_models.py:16-47, cs/video_level_models.py:14-19
and cs/validate.py:22-61: ``--flag value``, ``--flag=value``, booleans as a
separate token (``--frame_features True``, run_train.sh:6) or bare, list
strings with spaces (``--feature_names "rgb, audio"``); unknown flags are
tolerated (run_validate.sh:4 passes flags validate.py never defines).
"""
from __future__ import annotations

_DEFS = {}


def _define(name, default, typ, help_=""):
    _DEFS[name] = (default, typ, help_)


def _bool(v):
    if isinstance(v, bool):
        return v
    s = str(v).strip().lower()
    if s in ("true", "t", "1", "yes"):
        return True
    if s in ("false", "f", "0", "no"):
        return False
    raise ValueError("not a boolean: %r" % (v,))


# ---- cs/train.py:27-99 ------------------------------------------------------
_define("train_dir", "/tmp/yt8m_model/", str, "directory for checkpoints and logs")
_define("train_data_pattern", "", str, "glob of tf.SequenceExample records; '' or 'synthetic' = synthetic batches")
_define("feature_names", "rgb", str, "comma separated feature names")
_define("feature_sizes", "1024", str, "comma separated feature widths")
_define("frame_features", True, _bool)
_define("bagging", False, _bool)
_define("model", "HierarchicalLstmModel", str)
_define("start_new_model", False, _bool)
_define("batch_size", 1024, int)
_define("every_n", 1, int)
_define("label_loss", "CrossEntropyLoss", str)
_define("dropout", 0.5, float)
_define("regularization_penalty", 2.0, float)
_define("base_learning_rate", 0.001, float)
_define("learning_rate_decay", 1.0, float)
_define("learning_rate_decay_examples", 4000000.0, float)
_define("num_epochs", 10, int)
_define("num_readers", 4, int)
_define("optimizer", "AdamOptimizer", str)
_define("gpu", 0, int)
_define("clip_gradient_norm", 1.0, float)
_define("log_device_placement", False, _bool)
# ---- cs/frame_level_models.py:16-47 ---------------------------------------------
_define("iterations", 30, int)
_define("dbof_add_batch_norm", True, _bool)
_define("ppfs_normalize", False, _bool)
_define("sample_random_frames", True, _bool)
_define("dbof_cluster_size", 8192, int)
_define("dbof_hidden_size", 1024, int)
_define("dbof_pooling_method", "max", str)
_define("video_level_classifier_model", "MoeModel", str)
_define("lstm_cells", 1024, int)
_define("input_features", 1024, int)
_define("lstm_layers", 1, int)
_define("a_rate", "2", str)          # a *string* flag with an int default in the reference (:40)
_define("num_conv2d_layers", 4, int)
_define("filter_size", 10, int)
_define("max_num_frames", 300, int)
_define("num_inputs_to_lstm", 20, int)
_define("att_hid_size", 100, int)
# ---- cs/video_level_models.py:14-19 -----------------------------------------------
_define("moe_num_mixtures", 2, int)
_define("num_hidden_units", 1024, int)
# ---- cs/validate.py:22-61 (eval binaries) --------------------------------------------
_define("eval_data_pattern", "", str)
_define("run_once", False, _bool)
_define("top_k", 20, int)
# ---- additions of this build (not in the reference) --------------------------------
_define("max_steps", 0, int, "stop after this many iterations (0 = until the data ends)")
_define("synthetic_videos", 2048, int, "videos per epoch when train_data_pattern is synthetic")
_define("teacher_only", False, _bool, "HierarchicalLstmModel: train the teacher tower alone (BASELINE cfg 2; the reference "
        "always builds the student too, also at every_n=1)")
_define("precision", "bf16", str, "'bf16' (one bf16 MFMA product per forward contraction), 'high' (holds 1e-3 on logits at trained "
        "magnitudes: every forward product on IEEE f16 operands with the low-order halves of its weights - for the L1 level also of the input "
        "frames, for the MoE head of both operands - as OCP e4m3 operands on the MX-scaled MFMA behind the f16 stages of the same launch; the top layer of the L1 level "
        "contracts time-dithered f16 weight images instead (EVC_HIGH_DITHER_LAYERS, DESIGN.md 7); "
        "fixed power-of-two e4m3 scales: |x|, |h| <= 1, |W| < 4, head input |state| < 7, head weights |W| < 3.5 never clamp, larger values "
        "saturate at 448 and only lose their correction - engine.HLstmTower.fp8_saturation() counts them; the resolved layout depends on "
        "the EVC_HIGH_* environment and on the student's length and is logged / checkpointed as `precision_layout`) or "
        "'split' (split-bf16 operands, f32-operand accuracy, in every forward GEMM)")
_define("netvlad_cluster_size", 64, int, "NetVLADModel (extension): number of clusters")
_define("netvlad_hidden_size", 1024, int, "NetVLADModel (extension): width of the hidden layer after the aggregation")
_define("log_every", 1, int, "host metrics / logging period in iterations (the reference logs every step)")


class FlagValues(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.__dict__["_v"] = {k: d[0] for k, d in _DEFS.items()}
        self.__dict__["_unknown"] = []

    def __getattr__(self, k):
        try:
            return self.__dict__["_v"][k]
        except KeyError:
            raise AttributeError("unknown flag %s" % k)

    def __setattr__(self, k, v):
        if k not in _DEFS:
            raise AttributeError("unknown flag %s" % k)
        self.__dict__["_v"][k] = _DEFS[k][1](v)

    def flag_values_dict(self):
        return dict(self.__dict__["_v"])

    def parse(self, argv):
        """Parses reference-style argv (without the program name); returns the
        list of unknown tokens (ignored, like the reference's launchers rely on)."""
        i, unknown = 0, []
        while i < len(argv):
            tok = argv[i]
            i += 1
            if not tok.startswith("--"):
                unknown.append(tok)
                continue
            body = tok[2:]
            if "=" in body:
                name, val = body.split("=", 1)
            else:
                name, val = body, None
            if name.startswith("no") and name[2:] in _DEFS and _DEFS[name[2:]][1] is _bool and val is None:
                self.__dict__["_v"][name[2:]] = False
                continue
            if name not in _DEFS:
                unknown.append(tok)
                if val is None and i < len(argv) and not argv[i].startswith("--"):
                    unknown.append(argv[i])
                    i += 1
                continue
            typ = _DEFS[name][1]
            if val is None:
                if typ is _bool:
                    if i < len(argv) and not argv[i].startswith("--"):
                        try:
                            self.__dict__["_v"][name] = _bool(argv[i])
                            i += 1
                            continue
                        except ValueError:
                            pass
                    self.__dict__["_v"][name] = True
                    continue
                if i >= len(argv):
                    raise ValueError("flag --%s needs a value" % name)
                val = argv[i]
                i += 1
            self.__dict__["_v"][name] = typ(val)
        self.__dict__["_unknown"] = unknown
        return unknown


FLAGS = FlagValues()


def GetListOfFeatureNamesAndSizes(feature_names, feature_sizes):
    """cs/utils.py:127-148: split on ',' and strip; sizes to int."""
    names = [n.strip() for n in feature_names.split(",")]
    sizes = [int(s) for s in feature_sizes.split(",")]
    if len(names) != len(sizes):
        raise ValueError("length of the feature names (=%d) != length of feature sizes (=%d)" % (len(names), len(sizes)))
    return names, sizes
