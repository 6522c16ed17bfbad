"""CPU: the trained-like statistics generator (oracle/trained_like.py, test infrastructure) does what it claims."""


def test_trained_like_generator_spans_the_stated_ranges():
    """The generator really produces the statistics it claims (calibrated running_var over 5 decades, |gamma| up to 10,
    negative gammas, hot filters), and the detector still fires on a few percent of the pixels."""
    import numpy as np
    from oracle import trained_like as T
    from oracle import mp_oracle as O
    sd = T.trained_like_weights(11, dict(O.SHIPPED_MODEL_CONFIG), **T.SEVERITIES['wide+hot'])
    var = np.concatenate([v.numpy() for k, v in sd.items() if k.endswith('running_var') and not k.endswith('.5.running_var')])
    gam = np.concatenate([sd[k[:-len('running_var')] + 'weight'].numpy() for k in sd if k.endswith('running_var') and '.5.' not in k])
    assert var.min() < 3e-3 and var.max() > 30.0
    assert np.abs(gam).max() > 8.0 and np.abs(gam).min() < 0.13 and (gam < 0).mean() > 0.05


def test_structured_images_are_structured():
    """Piecewise-constant kinds hold large exactly-flat regions (the source of exact ties in the CPU probability map), all
    images stay inside [0, 1] and saturate somewhere."""
    import numpy as np
    from oracle import trained_like as T
    img = T.structured_images(3, 6, 96, 128).numpy()[:, 0]
    assert img.min() >= 0.0 and img.max() <= 1.0
    flat = [(np.diff(i, axis=1) == 0).mean() for i in img]
    assert flat[3] > 0.5 and flat[5] < 0.5          # kind 3: no noise; kind 5: sensor noise 0.05
    assert any((i == 0).any() or (i == 1).any() for i in img)


def test_fp64_oracle_bounds_the_fp32_oracle():
    """forward64 is the same restatement evaluated in double: on benign weights the fp32 oracle sits ~1e-5 from it, on the
    'wide+hot' weights two orders further -- the reason the GPU tolerance there is relative to ATen's own error."""
    from oracle import trained_like as T
    e = {}
    for sev in ('benign', 'wide+hot'):
        cfg, sd, img, r32, r64 = T.case(sev, 11, 1, 96, 128)
        e[sev] = float((r32['prob'].double() - r64['prob']).abs().max())
    assert e['benign'] < 5e-5
    assert e['wide+hot'] > 2 * e['benign']
