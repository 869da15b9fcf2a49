"""FLAGS: the reference's global flag set (main.py:33-80), same names and defaults, as a plain
namespace.  Models and trainers read the module-level FLAGS exactly like the reference reads
tf.app.flags.FLAGS (models/unet_acresnet.py:6-7, trainer/mfcctrainer.py:9-10)."""
import argparse

_DEFS = [
    ("mode", str, None), ("model", str, None), ("train_file", str, None), ("valid_file", str, None),
    ("test_file", str, None), ("exp_name", str, None), ("init_checkpoint", str, None),
    ("acoustic_init_checkpoint", str, None), ("audio_init_checkpoint", str, None),
    ("visual_init_checkpoint", str, None), ("restore_checkpoint", str, None), ("batch_size", int, 8),
    ("learning_rate", float, 0.001), ("latent_loss", float, 0.000001), ("display_freq", int, 1),
    ("num_epochs", int, 100), ("total_length", int, 30), ("sample_length", int, 1),
    ("number_of_crops", int, 30), ("buffer_size", int, 100), ("tensorboard", str, None),
    ("checkpoint_dir", str, None), ("temporal_pooling", int, 0), ("embedding", int, 0),
    ("margin", float, 0.2), ("block_size", int, 1), ("num_class", int, 128), ("datatype", str, "outdoor"),
    ("correspondence", int, 0), ("proxy", int, 0), ("encoder_type", str, "Video"), ("fusion", int, 0),
    ("moddrop", int, 0), ("l2", int, 0), ("project", int, 0), ("jointmvae", int, 0),
    ("onlyaudiovideo", int, 0), ("mfcc", int, 0), ("mfccmap", int, 0), ("num_skip_conn", int, 1),
    ("ae", int, 0), ("MSE", int, 1), ("huber_loss", int, 1),
    # not a reference flag: 1 = Trainer.train() runs the two-lane pipeline (frozen trunk of the next batch beside the
    # optimisation step of the current one; same numbers, the log line of an iteration one call later), 0 = one stream
    ("pipeline", int, 1),
]


class _Flags(object):
    def __init__(self):
        for name, _, default in _DEFS:
            setattr(self, name, default)

    def parse(self, argv=None):
        ap = argparse.ArgumentParser()
        for name, typ, default in _DEFS:
            ap.add_argument("--" + name, type=typ, default=default)
        ns, rest = ap.parse_known_args(argv)
        for name, _, _ in _DEFS:
            setattr(self, name, getattr(ns, name))
        return rest

    def as_dict(self):
        return dict((name, getattr(self, name)) for name, _, _ in _DEFS)


FLAGS = _Flags()
