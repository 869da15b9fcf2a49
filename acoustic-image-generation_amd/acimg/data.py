"""Loaders with the surface of the reference's (dataloader/outdoor_data_mfcc.py): `.data` (iterable of 6-tuples
(acoustic [n,36,48,12], mfcc [n,12], video [n,224,298,3], action one-hot, location one-hot, mfcc of the low-passed
audio [n,12])), `.num_samples`, `.total_batches` (:117,214,973-976).

* `SyntheticDataLoader`: seeded synthetic tensors at the reference's shapes and value ranges (every tensor
  entering the hot path is float32 in [0,1]; acoustic image and MFCC vector min-max normalised per sample).
* `TFRecordDataLoader` (round 3): the reference's ON-DISK format end to end - GZIP TFRecords of `SequenceExample`s
  (convert_data.py:247-279) through the native reader behind the C ABI (`acimg_gzip_inflate`, `acimg_tfrecord_index`,
  `acimg_sequence_example_decode`: `_parse_sequence`, :263-343, with its LR + UD flip), the audio front end on the
  device (`acimg_filtfilt` = `butter_lowpass_filter`, :558-575; `acimg_mfcc_frontend` = `_build_spectrograms_function`,
  :796-876, with `_normalize_mfcc`, :696-703), the per-frame maps of :634-703 on the host, unbatch to frames and batch
  (:99-104).  The tf.data machinery around it (parallel map, prefetch, shuffle buffer) is host plumbing the reference
  leaves to TensorFlow: here a plain Python generator (SURVEY §2 #16 keeps that pipeline out of scope)."""
import numpy as np
import torch


class SyntheticDataLoader(object):
    def __init__(self, num_samples, batch_size, num_actions=10, num_locations=61, seed=1234, device="cpu"):
        self.num_samples = int(num_samples)
        self.batch_size = int(batch_size)
        self.total_batches = -(-self.num_samples // self.batch_size)
        self.num_actions, self.num_locations = num_actions, num_locations
        self.seed = seed
        self.device = device
        self.data = self

    def _batch(self, n, seed):
        g = torch.Generator().manual_seed(seed)
        video = torch.rand(n, 224, 298, 3, generator=g)
        mfcc = torch.rand(n, 12, generator=g)
        mfcc = mfcc - mfcc.amin(1, keepdim=True)
        mfcc = mfcc / mfcc.amax(1, keepdim=True)
        ac = torch.rand(n, 36, 48, 12, generator=g)
        ac = ac - ac.amin((1, 2, 3), keepdim=True)
        ac = ac / ac.amax((1, 2, 3), keepdim=True)
        labels = torch.nn.functional.one_hot(torch.randint(0, self.num_actions, (n,), generator=g), self.num_actions)
        scen = torch.nn.functional.one_hot(torch.randint(0, self.num_locations, (n,), generator=g), self.num_locations)
        return ac, mfcc, video, labels.float(), scen.float(), mfcc.clone()

    def __iter__(self):
        left = self.num_samples
        i = 0
        while left > 0:
            n = min(self.batch_size, left)
            yield self._batch(n, self.seed + i)
            left -= n
            i += 1


class TFRecordDataLoader(object):
    """`ActionsDataLoader(txt_file, mode, batch_size, ..., embedding=1, normalize=..., shuffle=False)` of the reference for
    the MFCC path (FLAGS.mfcc): one record = one second = 12 frames; frames are unbatched and re-batched to
    `batch_size` (:99-104).  `files`: a list of TFRecord paths, or the path of a text file listing them (:214-236).
    device: where the audio front end runs (a GPU: there is no CPU fallback); tensors are returned on the host, like the
    reference's session.run results, and `Trainer._feed` copies them up."""

    def __init__(self, files, batch_size, num_actions=10, num_locations=61, device="cuda:0", frames_per_record=12,
                 compression_verify=True):
        from .frontend import FrontEnd
        if isinstance(files, str):
            with open(files) as f:
                files = [ln.strip() for ln in f if ln.strip()]
        self.files = list(files)
        self.batch_size = int(batch_size)
        self.num_actions, self.num_locations = int(num_actions), int(num_locations)
        self.frames = int(frames_per_record)
        self.verify = bool(compression_verify)
        self.device = torch.device(device)
        self.fe = FrontEnd(self.device)
        self.data = self
        self._num_samples = None

    # ---- :117, :973-976 ---------------------------------------------------------------------------------------------
    @property
    def num_samples(self):
        """frames in the data set (every record is `frames_per_record` frames): counted once from the record index"""
        if self._num_samples is None:
            from . import tfio
            self._num_samples = sum(len(tfio.read_tfrecord_native(p, verify=False)) for p in self.files) * self.frames
        return self._num_samples

    @property
    def total_batches(self):
        return -(-self.num_samples // self.batch_size)

    # ---- one record -> 12 frames (:263-343, :434-476, :558-575, :634-703) ---------------------------------------------
    def _record(self, rec):
        from . import tfio
        d = tfio.decode_sequence_example_native(rec)
        ai, sa, vi = d["audio_images"], d["audio_samples"], d["video_images"]
        n = vi.shape[0]
        if not (ai.shape[0] == n and sa.shape[0] == n and sa.shape[1] == 1024):
            raise ValueError("record with %d video / %d acoustic / %d audio steps" % (n, ai.shape[0], sa.shape[0]))
        # audio: raw frames -> device; low-passed copy (`filtered_wav`, :562) and the two MFCC vectors, each min-max
        # normalised per frame (`_normalize_mfcc`)
        frames = torch.from_numpy(np.ascontiguousarray(sa)).to(self.device)
        mfcc = self.fe._build_spectrograms_function(frames, normalize=True)
        low = self.fe.butter_lowpass_filter(frames)
        mfcc_low = self.fe._build_spectrograms_function(low, normalize=True)
        # acoustic images: per-frame min-max (`_normalize_acoustic_images_rescaled`, :672-679)
        a = ai.astype(np.float32)
        a = a - a.min(axis=(1, 2, 3), keepdims=True)
        a = a / a.max(axis=(1, 2, 3), keepdims=True)
        # video: float, channel order reversed, 1/255 (`_normalize_images_rescaled`, :649-655)
        v = vi[..., ::-1].astype(np.float32) * np.float32(1.0 / 255.0)
        act = np.zeros((n, self.num_actions), np.float32)
        act[:, d["action"]] = 1.0
        loc = np.zeros((n, self.num_locations), np.float32)
        loc[:, d["location"]] = 1.0
        return (torch.from_numpy(a), mfcc.cpu(), torch.from_numpy(np.ascontiguousarray(v)), torch.from_numpy(act),
                torch.from_numpy(loc), mfcc_low.cpu())

    def __iter__(self):
        from . import tfio
        pend, have = [], 0
        for path in self.files:
            for rec in tfio.read_tfrecord_native(path, verify=self.verify):
                if len(rec) == 0:
                    continue
                pend.append(self._record(rec))
                have += pend[-1][0].shape[0]
                while have >= self.batch_size:
                    cat = [torch.cat([p_[k] for p_ in pend], 0) for k in range(6)]
                    yield tuple(c[:self.batch_size] for c in cat)
                    rest = tuple(c[self.batch_size:] for c in cat)
                    have = rest[0].shape[0]
                    pend = [rest] if have else []
        if have:
            yield tuple(torch.cat([p_[k] for p_ in pend], 0) for k in range(6))
