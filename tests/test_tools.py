"""The profile-summary helpers under tools/ that run without a GPU."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
from _names import pretty  # noqa: E402


def test_kernel_names_with_bf16_template_arguments_are_decoded():
    """rocprofv3 (and binutils' c++filt) leave kernel names with __bf16 template arguments (Itanium `DF16b`) mangled; the summaries
    under profiles/ decode the argument kinds our kernels use."""
    assert pretty('_ZN12_GLOBAL__N_120conv3x3_wgrad_kernelILi64ELi3EDF16bDF16bEEvNS_9Wgrad3ArgsE') == 'conv3x3_wgrad_kernel<64, 3, bf16, bf16>'
    assert pretty('_ZN12_GLOBAL__N_119bn_bwd_apply_kernelIDF16bfDF16bEEvPKT0_P') == 'bn_bwd_apply_kernel<bf16, float, bf16>'
    assert pretty('_ZN12_GLOBAL__N_119wgrad_stream_kernelILi4ELi1ELi1ELi1ELi8ELb0EDF16bfEEvNS_9WgradArgsE') == \
        'wgrad_stream_kernel<4, 1, 1, 1, 8, false, bf16, float>'
    # names the profiler demangled itself pass through, without the parameter list and the anonymous namespace
    assert pretty('void (anonymous namespace)::gemm_kernel<64, 64, 4>((anonymous namespace)::GemmArgs)') == 'gemm_kernel<64, 64, 4>'
    assert pretty('adam_kernel(float*, float const*)') == 'adam_kernel'
    assert pretty('_ZN12_GLOBAL__N_111adam_kernelEPf') == 'adam_kernel'
