"""CPU oracle for the acoustic-image generation hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this package, and only as the checker.  The product (``acoustic-image-generation_amd/acimg``) never
imports it and has no CPU fallback.

What it is: a PyTorch-CPU (fp32 or fp64) + NumPy restatement of the reference's TensorFlow-1.x
graph for the path ``main.py --model UNet --embedding 1 --mfcc 1``:

    oracle/resnet50.py        models/vision.py:45-71, models/resnet50.py:75-125,128-225,253-276
    oracle/unet_acresnet.py   models/unet_acresnet.py:43-101,136-217 (+0skip/2skip variants)
    oracle/trainer.py         trainer/mfcctrainer.py:28-82 (loss + Adam), :411-442 (evaluate)
    oracle/frontend.py        dataloader/outdoor_data_mfcc.py:796-876,696-703; iouenergythreshold.py:294-323

PINNING STATUS
  * network path: **parity unpinned at the TensorFlow boundary.**  The arithmetic lives in
    TensorFlow >=1.14,<2 (tf.layers / tf.contrib.slim / tf.losses / tf.train.AdamOptimizer), which is
    not vendored in the reference, not installed here and not installable (no network); the
    reference ships no tests, golden vectors or fixtures.  The TF semantics restated here
    (SAME padding, conv2d_transpose VALID size, fused-BN moving variance, loss reductions, Adam
    epsilon placement, reduce_min/max tie gradients) are listed in SURVEY.md App. B.
  * audio front end + find_logen: **pinned** against the reference's own NumPy code, imported in the
    build container with TensorFlow stubbed out; vectors under tests/golden/ were produced by
    tests/golden/make_frontend_golden.py.
"""
