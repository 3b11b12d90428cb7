"""Reference switches this build does not implement must RAISE, not silently render the default behaviour
(VERDICT r2 'what's missing' #3): cfg.aggr in {dist, mindist} (inb_part_network_multiassign.py:240-251), knn_k != 4,
part_deform, tpose_viewdir False, use_knn False — in make/adopt and in Network.__init__.  (aggr = 'mean' and random_bg are built.)"""
import pytest

from invr import config
from invr.network import Network


BAD = [('aggr', 'median'), ('knn_k', 3), ('knn_k', 8), ('part_deform', True), ('tpose_viewdir', False), ('use_knn', False), ('use_amp', True)]


@pytest.mark.parametrize('key,val', BAD)
def test_network_rejects_unsupported_switch(key, val):
    cfg = config.make_cfg(table_log2=8, **{key: val})
    with pytest.raises(ValueError, match=key):
        Network(cfg=cfg)


@pytest.mark.parametrize('key,val', BAD)
def test_adopt_rejects_unsupported_switch(key, val):
    saved = dict(config.cfg)
    host = dict(config.DEFAULTS)
    host[key] = val
    try:
        with pytest.raises(ValueError, match=key):
            config.adopt(host)
    finally:
        config.set_cfg(config._node(saved))


def test_random_bg_is_built_on_the_fused_paths_only():
    # inb_renderer.py:72 hands cfg.random_bg to volume_rendering as render_weights' epsilon: built (fused forward / backward); the
    # op-by-op training graph composites with epsilon 0 and must refuse
    Network(cfg=config.make_cfg(table_log2=8, random_bg=True))
    with pytest.raises(ValueError, match='random_bg'):
        Network(cfg=config.make_cfg(table_log2=8, random_bg=True, train_fused=False))


def test_aggr_mean_is_built():
    Network(cfg=config.make_cfg(table_log2=8, aggr='mean'))
    Network(cfg=config.make_cfg(table_log2=8, aggr='mean', train_fused=False))
    for a in ('dist', 'mindist'):                      # round 5 (inb_part_network_multiassign.py:240-251)
        Network(cfg=config.make_cfg(table_log2=8, aggr=a))
        Network(cfg=config.make_cfg(table_log2=8, aggr=a, train_fused=False))


def test_defaults_and_ignored_keys_pass():
    saved = dict(config.cfg)
    host = dict(config.DEFAULTS)
    host['N_importance'] = 128          # set by inb_377.yaml, read by nothing in the reference's lib/: ignored as there
    try:
        c = config.adopt(host)
        assert c.aggr == '' and c.knn_k == 4
        Network(cfg=config.make_cfg(table_log2=8))
    finally:
        config.set_cfg(config._node(saved))
