"""MJCF-subset compiler and bounds (host logic; CPU)."""

import numpy as np
import pytest

from conftest import REFERENCE
from stac_mjx_amd.mjcf import (JNT_BALL, JNT_FREE, JNT_HINGE, JNT_SLIDE, MjcfError, ModelTables,
                               align_joint_dims, compile_mjcf)


def test_toy_model_layout(toy_tables):
    t = toy_tables
    assert (t.nbody, t.njnt, t.nq, t.nsite) == (4, 3, 3, 3)
    assert t.body_parentid.tolist() == [0, 0, 1, 2]
    assert t.body_names == ["world", "b1", "b2", "b3"]
    np.testing.assert_array_equal(t.jnt_axis, np.eye(3, dtype=np.float32)[[2, 0, 1]])
    np.testing.assert_allclose(t.site_pos[1], [0.4, 0.5, 0.6])
    assert t.jnt_type.tolist() == [JNT_HINGE] * 3


def test_align_joint_dims_reference_vectors():
    """Restates tests/unit/test_controller.py:27-89 (exact lb/ub/part_names)."""
    types = [JNT_FREE, JNT_HINGE, JNT_BALL, JNT_SLIDE]
    ranges = [[0.0, 0.0], [-0.1, 0.1], [0.0, 1.0], [-0.5, 0.5]]
    names = ["root", "hingejoint", "balljoint", "slidejoint"]
    lb, ub, part_names = align_joint_dims(types, ranges, names)
    inf = np.inf
    np.testing.assert_array_equal(lb, np.array([-inf] * 3 + [-1.0] * 4 + [-0.1] + [0.0] * 4 + [-0.5], np.float32))
    np.testing.assert_array_equal(ub, np.array([inf] * 3 + [1.0] * 4 + [0.1] + [1.0] * 4 + [0.5], np.float32))
    assert part_names == ["root"] * 7 + ["hingejoint"] + ["balljoint"] * 4 + ["slidejoint"]


def test_unconstrained_hinge_and_lb_min_zero():
    lb, ub, _ = align_joint_dims([JNT_HINGE, JNT_HINGE], [[0, 0], [0.2, 0.9]], ["a", "b"])
    np.testing.assert_allclose(lb, [-2 * np.pi, 0.0], rtol=1e-7)  # lb = min(lb, 0): stac.py:88
    np.testing.assert_allclose(ub, [2 * np.pi, 0.9], rtol=1e-7)


def test_defaults_childclass_degrees_and_euler():
    xml = """
    <mujoco>
      <default>
        <joint axis="0 1 0" range="-90 90"/>
        <default class="a"><joint axis="2 0 0" pos="0.1 0 0"/><default class="b"><joint range="-45 45"/></default></default>
      </default>
      <worldbody>
        <body name="r" pos="0 0 1" euler="0 0 90">
          <freejoint name="root"/>
          <body name="c1" pos="1 0 0" childclass="a">
            <joint name="j1"/>
            <body name="c2" pos="0 2 0" quat="2 0 0 0">
              <joint name="j2" class="b"/>
              <joint name="j3" type="slide" axis="0 0 3" range="-1 1" class="main"/>
            </body>
          </body>
        </body>
      </worldbody>
    </mujoco>"""
    t = compile_mjcf(xml, from_string=True)
    assert (t.nbody, t.njnt, t.nq) == (4, 4, 10)
    assert t.jnt_type.tolist() == [JNT_FREE, JNT_HINGE, JNT_HINGE, JNT_SLIDE]
    np.testing.assert_allclose(t.jnt_axis[1], [1, 0, 0])  # class a via childclass, normalised
    np.testing.assert_allclose(t.jnt_pos[2], [0.1, 0, 0])  # class b inherits a
    np.testing.assert_allclose(t.jnt_range[1], np.deg2rad([-90, 90]), rtol=1e-6)  # degrees by default
    np.testing.assert_allclose(t.jnt_range[2], np.deg2rad([-45, 45]), rtol=1e-6)
    np.testing.assert_allclose(t.jnt_range[3], [-1, 1])  # slide range is not an angle
    np.testing.assert_allclose(t.jnt_axis[3], [0, 0, 1])
    np.testing.assert_allclose(t.body_quat[1], [np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)], atol=1e-7)
    np.testing.assert_allclose(t.body_quat[3], [1, 0, 0, 0])  # quat normalised
    np.testing.assert_allclose(t.qpos0[:7], [0, 0, 1, np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)], atol=1e-7)


def test_scale_rule_matches_dm_scale_spec():
    """rescale.py:21-45: only body pos strictly below the first top-level body is scaled."""
    xml = """
    <mujoco><worldbody>
      <body name="a" pos="1 0 0"><joint name="ja" pos="0.5 0 0"/>
        <body name="a1" pos="0 2 0"><joint name="ja1" pos="0 0.5 0"/><site name="s" pos="0 0 1"/></body></body>
      <body name="b" pos="3 0 0"><body name="b1" pos="0 4 0"/></body>
    </worldbody></mujoco>"""
    t = compile_mjcf(xml, from_string=True, scale=0.5)
    np.testing.assert_allclose(t.body_pos[1:], [[1, 0, 0], [0, 1, 0], [3, 0, 0], [0, 4, 0]])
    np.testing.assert_allclose(t.jnt_pos, [[0.5, 0, 0], [0, 0.5, 0]])  # joint pos never scaled
    np.testing.assert_allclose(t.site_pos, [[0, 0, 1]])  # site pos never scaled
    tl = compile_mjcf(xml, from_string=True, scale=0.5, legacy_joint_pos_scale=True)
    np.testing.assert_allclose(tl.jnt_pos, [[0.25, 0, 0], [0, 0.25, 0]])


def test_unsupported_constructs_raise():
    with pytest.raises(MjcfError):
        compile_mjcf("<mujoco><include file='x.xml'/><worldbody/></mujoco>", from_string=True)
    with pytest.raises(MjcfError):
        compile_mjcf("<mujoco><worldbody><body><joint type='screw'/></body></worldbody></mujoco>", from_string=True)


def test_tables_roundtrip(tmp_path, rodent_setup):
    p = tmp_path / "t.npz"
    rodent_setup.tables.save(p)
    t2 = ModelTables.load(p)
    for k, v in rodent_setup.tables.__dict__.items():
        if isinstance(v, np.ndarray):
            np.testing.assert_array_equal(v, getattr(t2, k))
        else:
            assert v == getattr(t2, k)


def test_rodent_facts(rodent_setup, demo_viz):
    """SURVEY.md F9 + names pinned by the stored reference output."""
    fs = rodent_setup
    t = fs.tables
    assert (t.nbody, t.njnt, t.nq, t.nsite) == (67, 68, 74, 23)
    assert t.body_names == demo_viz["names_xpos"].tolist()
    assert fs.part_names == demo_viz["names_qpos"].tolist()
    assert fs.kp_names == demo_viz["kp_names"].tolist()
    assert fs.part_masks.sum(1).tolist() == [11, 11, 6, 6, 7]
    assert fs.trunk_kps.sum() == 8 and fs.root_kp_idx == 18 and fs.root_dims == 7
    assert fs.is_regularized.sum() == 15
    assert np.all(fs.lb[:3] == -np.inf) and np.all(fs.lb[3:7] == -1) and np.all(fs.lb <= 0)


def test_fly_facts(fly_setup):
    t = fly_setup.tables
    assert (t.nbody, t.nq, t.nsite) == (68, 43, 30)
    assert fly_setup.root_kp_idx == -1 and not fly_setup.do_root_opt
    assert fly_setup.part_masks.shape == (6, 43)


@pytest.mark.skipif(not REFERENCE.exists(), reason="reference checkout not present")
def test_compile_matches_committed_fixture(rodent_setup, rodent_cfg):
    from stac_mjx_amd.fit_model import build_fit_setup

    fs = build_fit_setup(REFERENCE / "models" / "rodent.xml", rodent_cfg, rodent_setup.kp_names)
    for k, v in rodent_setup.tables.__dict__.items():
        if isinstance(v, np.ndarray):
            np.testing.assert_array_equal(v, getattr(fs.tables, k), err_msg=k)


@pytest.mark.skipif(not REFERENCE.exists(), reason="reference checkout not present")
@pytest.mark.parametrize("rel", ["models/synth_model.xml", "models/mouse/mouse_with_meshes.xml",
                                 "models/fruitfly/fruitfly_force_ball.xml", "models/celegans/celegans.xml"])
def test_other_reference_models_compile(rel):
    t = compile_mjcf(REFERENCE / rel)
    assert t.nq > 0 and t.nbody > 1
    assert t.nq == sum({JNT_FREE: 7, JNT_BALL: 4}.get(int(x), 1) for x in t.jnt_type)
