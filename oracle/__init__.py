"""
oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement (NumPy, float64 or float32) of the arithmetic on the DeepPrior++
hot path: the layer library, the ResNet / PoseRegNet graphs, the train step
(sum-squared-error loss, exact reverse-mode gradients, the reference's ADAM),
and the crop augmentation (cv2-NEAREST warps restated from OpenCV's published
algorithm).  Every function cites the reference file:line it follows.

PARITY UNPINNED: the reference ships no tests, golden vectors or pre-trained
weights, and its numerical path (Theano 0.9 + cv2 2.4, Python 2.7) cannot be
imported or executed in this environment.  What CAN be
imported from /root/reference (the *LayerParams shape arithmetic, the 2-D point
transforms, `chunks`) is used by tests/golden/make_golden.py to generate the
committed fixtures that pin those parts; the rest is pinned by an independent
cross-check against torch CPU autograd (oracle/torch_ref.py) in float64.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product path (deep-prior-pp_amd/) never does.
"""
