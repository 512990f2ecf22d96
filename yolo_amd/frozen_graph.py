"""Named-tensor view of a compiled network: the terminal nodes the reference freezes into its protobuf
(convert_to_pb.py:11,69-70,78) and looks up again in predict_from_pb.py:221-227 --

    input_image   float32 [B,S,S,3] in [0,1]      (placeholder, convert_to_pb.py:69)
    input_shape   int32 [2] = source (h, w)        (placeholder, convert_to_pb.py:70)
    scale_1..3    head tensors [B,g,g,3*(5+C)], coarse first (tf.identity of yolo()'s outputs, :78)

No protobuf is read or written: the "graph" is a Network built from a Darknet cfg + .weights, and `run`
launches its HIP plan.  The shim exists so code written against the frozen graph's node names
(`graph.get_tensor_by_name('scale_1:0')`, `sess.run(fetches, feed_dict)`) has something to bind to.
"""
import numpy as np
import torch

from .build_model import Network, OutputNodes

terminal_node_names = ["input_image", "input_shape", "scale_1", "scale_2", "scale_3"]    # convert_to_pb.py:11


class NamedTensor(object):
    """What get_tensor_by_name returns: a name to put in `fetches` or `feed_dict`."""

    def __init__(self, graph, name):
        self.graph, self.name = graph, name

    def __repr__(self):
        return "<NamedTensor %s:0>" % self.name

    def __hash__(self):
        return hash(self.name)

    def __eq__(self, other):
        return isinstance(other, NamedTensor) and other.name == self.name and other.graph is self.graph


def _node(name):
    name = name.name if isinstance(name, NamedTensor) else str(name)
    base, _, port = name.partition(":")
    if port not in ("", "0"):
        raise KeyError("tensor %r: the terminal nodes have a single output (':0')" % name)
    return base


class FrozenGraph(object):
    def __init__(self, network):
        if isinstance(network, OutputNodes):
            network = network.net
        if not isinstance(network, Network):
            raise TypeError("FrozenGraph wraps a yolo_amd Network (or the OutputNodes of yolo())")
        self.net = network
        self.outputs = ["scale_%d" % (k + 1) for k in range(len(network.out_index))]
        self.image_shape = None            # last value fed to input_shape

    def node_names(self):
        return ["input_image", "input_shape"] + self.outputs

    def get_tensor_by_name(self, name):
        base = _node(name)
        if base not in self.node_names():
            raise KeyError("The name %r refers to a Tensor which does not exist (nodes: %s)"
                           % (name, ", ".join(self.node_names())))
        return NamedTensor(self, base)

    def run(self, fetches, feed_dict):
        """sess.run(fetches, feed_dict) over the terminal nodes (predict_from_pb.py:250-251): feeds
        `input_image` (numpy or torch, [B,S,S,3] or [S,S,3]) and optionally `input_shape`; returns the
        fetched head tensors (torch, on the GPU) in fetch order -- a single tensor for a single fetch."""
        feeds = {_node(k): v for k, v in feed_dict.items()}
        unknown = [k for k in feeds if k not in ("input_image", "input_shape")]
        if unknown:
            raise KeyError("cannot feed %s: only input_image and input_shape are placeholders" % unknown)
        single = not isinstance(fetches, (list, tuple))
        names = [_node(f) for f in ([fetches] if single else fetches)]
        for n in names:
            if n not in self.node_names():
                raise KeyError("The name %r refers to a Tensor which does not exist" % n)
        if "input_shape" in feeds:
            shape = np.asarray(feeds["input_shape"]).astype(np.int32).reshape(-1)
            if shape.shape != (2,):
                raise ValueError("input_shape must have shape [2] (convert_to_pb.py:70)")
            self.image_shape = shape
        heads = None
        if any(n in self.outputs for n in names):
            if "input_image" not in feeds:
                raise ValueError("You must feed a value for placeholder tensor 'input_image'")
            img = feeds["input_image"]
            if not isinstance(img, torch.Tensor):
                img = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32))
            if img.dim() == 3:
                img = img.unsqueeze(0)
            heads = self.net.forward(img.to(self.net.device, dtype=torch.float32))
        out = []
        for n in names:
            if n in self.outputs:
                out.append(heads[self.outputs.index(n)])
            elif n == "input_image":
                out.append(feeds.get("input_image"))
            else:
                if self.image_shape is None:
                    raise ValueError("You must feed a value for placeholder tensor 'input_shape'")
                out.append(self.image_shape)
        return out[0] if single else out
