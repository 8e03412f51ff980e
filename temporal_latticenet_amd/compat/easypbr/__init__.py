"""harness stand-in: inert versions of the easy_pbr classes the reference's drivers construct unconditionally
(train_ln.py:101-103, dataloader/kitti_dataloader.py:295).  GUI / IO, not part of the hot path."""


class Mesh:
    def __init__(self, *a, **k):
        self.V = self.C = self.I = self.L_gt = self.L_pred = None
        self.m_vis = type("Vis", (), {})()
        self.m_label_mngr = None

    def __getattr__(self, name):
        return lambda *a, **k: None


class LabelMngr:
    def __init__(self, *a, **k):
        self.args = a

    def __getattr__(self, name):
        return lambda *a, **k: None


class Scene:
    @staticmethod
    def show(*a, **k):
        return None


class Viewer:
    @staticmethod
    def create(*a, **k):
        return Viewer()

    def update(self, *a, **k):
        return None
