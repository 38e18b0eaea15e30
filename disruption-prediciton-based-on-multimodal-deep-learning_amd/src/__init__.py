"""MI355X-native drop-in for the hot path of the reference's ``src`` package.

Put this directory's PARENT on ``PYTHONPATH`` and the reference's training scripts import
``src.models.*``, ``src.loss``, ``src.train``, ``src.GradientBlending`` and ``src.distributed`` from here.
All compute runs in the gfx950 shared library ``csrc/libmi355x_disrupt.so`` (C ABI: include/mi355x_disrupt.h);
there is no CPU fallback -- a missing library or a CPU tensor raises.
"""
