"""Host-side mirror of the reference's ``XMLManager`` (one class serves all four robots).

Reference: ``envs/flamingo_light_v1/manager/xml_manager.py:16-122`` and its three sibling
copies (they differ only in the body lists, see ``robots.py``).  The reference rewrites
the MJCF on disk next to its sources on every construction; here
the same seven edits are applied to an in-memory element tree and nothing is written.

Step 3 (mass noise + load) draws ``np.random.uniform`` once per listed body *in document
order*, exactly like the reference, when ``rng is None`` — that is what the golden
vectors in ``tests/golden/xml_*.json`` pin.  For batches the caller passes ``num_envs``
and a ``numpy.random.Generator`` and gets ``[num_envs, nbody]`` mass arrays instead.
"""
from __future__ import annotations

import copy
import os
import xml.etree.ElementTree as ET
from typing import Dict, Optional

import numpy as np

from .robots import ROBOTS

ASSET_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")


class XMLManager:
    def __init__(self, config: dict, model_path: Optional[str] = None):
        self.config = config
        self.env_id = config["env"]["id"]
        if self.env_id not in ROBOTS:
            raise NameError(f"Please select a valid environment id. Received '{self.env_id}'.")
        self.robot = ROBOTS[self.env_id]
        self.body_components = list(self.robot["mass_bodies"])
        self.precision_attr_map = config["random_table"]["precision"]
        self.model_path = model_path or os.path.join(ASSET_DIR, self.env_id, self.robot["xml"])

    def get_model_tree(self) -> ET.Element:
        """Steps 1, 2, 4, 5, 6 of the reference (mass noise, step 3, is :meth:`draw_masses`)."""
        root = copy.deepcopy(ET.parse(self.model_path).getroot())
        cfg = self.config

        # 1. terrain (xml_manager.py:21-32)
        terrain = cfg["env"]["terrain"]
        for geom in root.findall(".//geom"):
            if geom.attrib.get("name") == "ground":
                if terrain == "flat":
                    geom.attrib["type"] = "plane"
                    geom.attrib.pop("hfield", None)
                    geom.attrib["size"] = "100 100 0.1"
                else:
                    geom.attrib["type"] = "hfield"
                    geom.attrib["hfield"] = terrain

        # 2. precision (:34-41)
        level = cfg["random"]["precision"]
        if level in self.precision_attr_map:
            option = root.find("option")
            if option is not None:
                option.attrib["timestep"] = str(self.precision_attr_map[level]["timestep"])
                option.attrib["iterations"] = str(self.precision_attr_map[level]["iterations"])

        # 4. wheel / foot geom friction, only where the attribute is written on the geom (:57-66)
        fr = (f"{cfg['random']['sliding_friction']} {cfg['random']['torsional_friction']} "
              f"{cfg['random']['rolling_friction']}")
        for body in root.findall(".//body"):
            if body.attrib.get("name") in self.robot["friction_bodies"]:
                for geom in body.findall("geom"):
                    if "friction" in geom.attrib:
                        geom.attrib["friction"] = fr

        # 5. ground friction (:68-75)
        for geom in root.findall(".//geom"):
            if geom.attrib.get("name") == "ground" and "friction" in geom.attrib:
                geom.attrib["friction"] = fr

        # 6. frictionloss on the default classes named joints / wheels (:77-87)
        for default in root.findall(".//default"):
            if default.attrib.get("class") in ("joints", "wheels"):
                for joint in default.findall("joint"):
                    if "frictionloss" in joint.attrib:
                        joint.attrib["frictionloss"] = str(cfg["random"]["friction_loss"])

        # 7. height-map marker sites are visual only (:89-118); the engine samples the terrain directly.
        return root

    def nominal_masses(self, root: ET.Element) -> Dict[str, float]:
        out = {}
        for body in root.findall(".//body"):
            for inertial in body.findall("inertial"):
                if "mass" in inertial.attrib:
                    out[body.attrib.get("name")] = float(inertial.attrib["mass"])
        return out

    def draw_masses(self, root: ET.Element, num_envs: int = 1,
                    rng: Optional[np.random.Generator] = None) -> Dict[str, np.ndarray]:
        """Step 3 (:43-55): ``mass += U(-m k, +m k)`` per listed body, base also ``+= load``.

        Returns ``{body name: mass[num_envs]}`` for the listed bodies, visiting them in
        document order.  With ``rng is None`` the legacy global ``np.random`` stream is used
        (one draw per body, ``num_envs`` must be 1) — same stream as the reference.
        """
        k = self.config["random"]["mass_noise"]
        load = self.config["random"]["load"]
        out: Dict[str, np.ndarray] = {}
        for body in root.findall(".//body"):
            name = body.attrib.get("name")
            if name in self.body_components:
                for inertial in body.findall("inertial"):
                    if "mass" in inertial.attrib:
                        m0 = float(inertial.attrib["mass"])
                        if rng is None:
                            if num_envs != 1:
                                raise ValueError("the legacy np.random stream serves one env; pass rng for batches")
                            noise = np.array([np.random.uniform(-m0 * k, m0 * k)])
                        else:
                            noise = rng.uniform(-m0 * k, m0 * k, size=num_envs)
                        m = m0 + noise
                        if name == self.robot["base_body"]:
                            m = m + load
                        out[name] = m
        return out
