"""Install the HIP backend under a live ``tsim`` (true drop-in for users who have it).

``tsim.sampler.sample_program`` is a module global resolved at call time by every call site
(reference src/tsim/sampler.py:274,400,484), and it is the function the reference's own tests
replace.  ``install()`` rebinds it (and ``tsim.compile.evaluate.evaluate`` for
``CompiledStateProbs``) to the HIP-backed implementations.
"""

from __future__ import annotations


def install(*, patch_evaluate: bool = False):
    """Rebind ``tsim.sampler.sample_program`` to the HIP backend; returns the previous function."""
    try:
        import tsim.sampler as ref_sampler  # type: ignore
    except Exception as exc:  # pragma: no cover - tsim is not installable in this image
        raise RuntimeError(
            "tsim is not importable here; use tsim_amd.sampler with an exported .npz program instead"
        ) from exc
    from . import backend

    previous = ref_sampler.sample_program

    def sample_program(program, f_params, key):
        return backend.sample_program(program, f_params, key)

    ref_sampler.sample_program = sample_program
    if patch_evaluate:
        import tsim.sampler as s  # the name `evaluate` imported into tsim.sampler (sampler.py:15)

        s.evaluate = backend.evaluate
    return previous


def uninstall(previous) -> None:
    import tsim.sampler as ref_sampler  # type: ignore

    ref_sampler.sample_program = previous
