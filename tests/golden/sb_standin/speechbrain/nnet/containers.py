"""Stand-in for speechbrain.nnet.containers (Sequential with input-shape inference, ModuleList)."""
import inspect
import torch


class Sequential(torch.nn.ModuleDict):
    def __init__(self, *layers, input_shape=None, **named_layers):
        super().__init__()
        if not layers and input_shape is None and not named_layers:
            raise ValueError("Must pass either layers or input shape")
        self.length_layers = []
        self.input_shape = input_shape
        if input_shape and None in input_shape:
            self.input_shape = list(input_shape)
            for i, dim in enumerate(self.input_shape):
                if i == 0 and dim is None:
                    dim = 1
                self.input_shape[i] = dim or 256
        for layer in layers:
            self.append(layer)
        for name, layer in named_layers.items():
            self.append(layer, layer_name=name)

    def append(self, layer, *args, layer_name=None, **kwargs):
        if layer_name is None:
            layer_name = str(len(self))
        elif layer_name in self:
            index = 0
            while f"{layer_name}_{index}" in self:
                index += 1
            layer_name = f"{layer_name}_{index}"
        if self.input_shape:
            argspec = inspect.getfullargspec(layer)
            if "input_shape" in argspec.args + argspec.kwonlyargs:
                kwargs["input_shape"] = self.get_output_shape()
        try:
            self.add_module(layer_name, layer(*args, **kwargs))
        except TypeError:
            self.add_module(layer_name, layer)

    def get_output_shape(self):
        with torch.no_grad():
            dummy = torch.zeros(self.input_shape)
            return self(dummy).shape

    def forward(self, x):
        for layer in self.values():
            x = layer(x)
            if isinstance(x, tuple):
                x = x[0]
        return x


class ModuleList(torch.nn.Module):
    def __init__(self, *layers):
        super().__init__()
        self.layers = torch.nn.ModuleList(layers)

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
            if isinstance(x, tuple):
                x = x[0]
        return x

    def append(self, module):
        self.layers.append(module)
